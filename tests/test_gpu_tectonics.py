"""Tectonics through the C ABI on the GPU: bit-exact against the llvmpipe goldens on power-of-two
grids and against the oracle elsewhere (every pass is a GL_NEAREST gather: parity is counted in differing texels)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def _bad(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return int((np.any(a != b, axis=-1) if a.ndim == 3 else (a != b)).sum())


@pytest.mark.parametrize("name", ["globe_64x32_s70", "globe_128x64_s10", "belt_128x64_s12", "region_128x64_s12"])
def test_tectonics_golden(dev, name):
    capi, ctx = dev
    G = load_golden("tectonics_" + name)
    H, W = G["img"].shape
    t = capi.Tectonics(ctx, capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"])), G["img"])
    t.fold()
    assert _bad(t.map(), G["fold1"]) == 0
    t.spread()
    assert _bad(t.map(), G["spread1"]) == 0
    t.unfold()
    t.step(int(G["steps"]) - 1)
    assert _bad(t.terrain(), G["terrain"]) == 0
    p = t.plates()
    assert _bad(p[0], G["plate0"]) == 0 and _bad(p[1], G["plate1"]) == 0
    t.close()


@pytest.mark.parametrize("W,H,steps,coords", [(256, 128, 7, None), (100, 50, 9, None), (512, 256, 3, None),
                                              (192, 64, 5, (-1.0, 1.2, -np.pi, np.pi)),
                                              (128, 128, 4, None), (256, 64, 4, None), (96, 40, 4, None)])
def test_tectonics_vs_oracle(dev, oracle, W, H, steps, coords):
    """other sizes (incl. non-power-of-two, where oracle and kernel share the exact st = (x+0.5)/W) and extents"""
    capi, ctx = dev
    from demiurge_amd import synth
    img = (synth.terrain(W, H) * 5 - 0.5).astype(np.float32)
    c = capi.FULL_GLOBE if coords is None else tuple(np.float32(v) for v in coords)
    ref_terrain, ref_plates = oracle.tectonics(oracle.grid(W, H, c), img, steps)
    t = capi.Tectonics(ctx, capi.make_grid(W, H, c), img)
    t.step(steps)
    assert _bad(t.terrain(), ref_terrain) == 0
    for a, b in zip(t.plates(), ref_plates):
        assert _bad(a, b) == 0
    t.close()


def test_tectonics_custom_velocity(dev, oracle):
    """Plate::updateRotationBy with an oblique axis: exercises every entry of the rotation matrices"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 128, 64
    img = (synth.terrain(W, H) * 4).astype(np.float32)
    extra = [(0, 0.02, (0.3, -0.5, 0.8)), (1, -0.015, (0.1, 0.9, 0.2))]
    ref_terrain, ref_plates = oracle.tectonics(oracle.grid(W, H), img, 8, extra_rotation=extra)
    t = capi.Tectonics(ctx, capi.make_grid(W, H), img)
    for k, theta, axis in extra:
        t.update_rotation(k, theta, axis)
    t.step(8)
    assert _bad(t.terrain(), ref_terrain) == 0
    for a, b in zip(t.plates(), ref_plates):
        assert _bad(a, b) == 0
    t.close()


def test_tectonics_large_properties(dev):
    """4096x2048 (size-independent properties): after oceanSpreading every map texel belongs to a plate, the flag
    channel is 0/1, plate storage cycles W x H -> W x W with period 3, and the run is deterministic"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 4096, 2048
    img = (synth.terrain(W, H) * 3).astype(np.float32)

    def run():
        t = capi.Tectonics(ctx, capi.make_grid(W, H), img)
        sizes = []
        for _ in range(4):
            t.fold(); t.spread(); t.unfold()
            sizes.append(tuple(p.shape[:2] for p in t.plates()))
        t.fold(); t.spread()
        m = t.map()
        t.render()
        ter = t.terrain()
        t.close()
        return sizes, m, ter

    sizes, m, ter = run()
    assert sizes[0] == ((W, W), (H, W)) and sizes[1] == ((H, W), (W, W)) and sizes[2] == ((H, W), (H, W)) and sizes[3] == sizes[0]
    assert set(np.unique(m[..., 0])) <= {1.0, 2.0}
    assert set(np.unique(m[..., 2])) <= {0.0, 1.0}
    assert set(np.unique(ter)) <= {1.0, 2.0}
    assert np.array_equal(ter, m[1::2, :, 0])                     # render = NEAREST resample of the W x W map
    sizes2, m2, ter2 = run()
    assert np.array_equal(m, m2) and np.array_equal(ter, ter2)


def test_tectonics_argument_errors(dev):
    capi, ctx = dev
    with pytest.raises(capi.DmgError):                                    # plates are W x H inside W x W maps: H <= W
        capi.Tectonics(ctx, capi.make_grid(32, 64), np.zeros((64, 32), np.float32))
    t = capi.Tectonics(ctx, capi.make_grid(64, 32), np.zeros((32, 64), np.float32))
    with pytest.raises(capi.DmgError):
        t.update_rotation(2, 0.1, (1, 0, 0))                              # the reference has two plates
    t.update_rotation(0, 0.01, (1, 0, 0))                                 # plate 0: -0.01 + 0.01 = no motion at all
    with pytest.raises(capi.DmgError):
        t.step(1)                                                         # glm::normalize(0) is NaN in the reference
    t.close()
