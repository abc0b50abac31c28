"""GradientNoise.  CPU: the oracle restatement reproduces the reference's GLSL on llvmpipe bit for
bit for all seven modes (power-of-two grids) and its MT19937 seed offsets.  GPU: the HIP kernels through the C ABI,
bit-exact against the same goldens and against the oracle at other sizes and parameters."""
import numpy as np
import pytest

from conftest import load_golden, relerr

MODES = ["DEFAULT", "RIDGED", "BILLOWY", "IQNoise", "SWISS", "Jordan", "Plateaus"]
KEYS_B = ("scale", "lacunarity", "persistence", "min", "max", "warp")


def _variants(G):
    B = dict(zip(KEYS_B, (float(v) for v in G["params_B"])), octaves=int(G["octaves_B"]), seed=int(G["seed_B"]))
    return [("region_DEFAULT_B", "DEFAULT", "replace", None, B), ("region_SWISS_B", "SWISS", "replace", None, B),
            ("region_sel_DEFAULT_replace", "DEFAULT", "replace", G["sel"], dict(seed=1)),
            ("region_sel_DEFAULT_add", "DEFAULT", "add", G["sel"], dict(seed=1)),
            ("region_sel_Jordan_replace", "Jordan", "replace", G["sel"], dict(seed=123456))]


def test_oracle_seed_offsets(oracle):
    """MT19937 + generate_canonical<float> restated == the reference's own statements run on libstdc++"""
    G = load_golden("noise_128x64_variants")
    for s, ref in zip(G["seeds"], G["seed_offsets"]):
        assert np.array_equal(oracle.noise_seed_offset(int(s)), ref), s
    # cross-check against numpy's independent MT19937: first three raw 32-bit outputs
    rs = np.random.RandomState(12345)
    raw = rs.randint(0, 2 ** 32, size=3, dtype=np.uint64).astype(np.float32) / np.float32(4294967296.0) * np.float32(10000.0)
    assert np.array_equal(oracle.noise_seed_offset(12345), raw.astype(np.float32))


@pytest.mark.parametrize("mode", MODES)
def test_oracle_noise_modes_golden(oracle, mode):
    G = load_golden("noise_128x64_modes")
    g = oracle.grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    out = oracle.gradient_noise(g, G["img"], mode, "replace", seed=int(G["seed"]), warp=float(G["warp"]))
    assert np.array_equal(out, G["out_" + mode])


def test_oracle_noise_variants_golden(oracle):
    """regional extent, non-default parameters, fractional selection, add blend"""
    G = load_golden("noise_128x64_variants")
    g = oracle.grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    for key, mode, blend, sel, prm in _variants(G):
        assert np.array_equal(oracle.gradient_noise(g, G["img"], mode, blend, sel=sel, **prm), G[key]), key


def test_oracle_noise_non_pow2(oracle):
    """100x50: llvmpipe interpolates st a few 1e-8 off (x+0.5)/W and the shader quantises the position to ~1e-3
    (seed offsets up to 1e4 in fp32), so a fraction of texels lands one quantum away: bounded, not bit-exact"""
    G = load_golden("noise_100x50_default")
    g = oracle.grid(100, 50)
    out = oracle.gradient_noise(g, G["img"], "DEFAULT", seed=int(G["seed"]))
    assert (out != G["out_DEFAULT"]).mean() < 0.05 and relerr(out, G["out_DEFAULT"]) < 5e-3
    # ... and with st as llvmpipe's rasteriser interpolates it (tools/st_probe.py) every texel is the reference's
    oracle.set_raster_st(True)
    try:
        assert np.array_equal(oracle.gradient_noise(g, G["img"], "DEFAULT", seed=int(G["seed"])), G["out_DEFAULT"])
    finally:
        oracle.set_raster_st(False)


def test_oracle_blend_modes(oracle):
    """filter::blendMode(): the five blends the goldens do not cover, against their one-line definitions"""
    g = oracle.grid(64, 32)
    rng = np.random.default_rng(0)
    old = rng.normal(size=(32, 64)).astype(np.float32)
    new = oracle.gradient_noise(g, np.zeros_like(old), "DEFAULT", seed=7)
    for blend, fn in dict(subtract=lambda o, n: np.maximum(o - n, np.float32(0)), multiply=lambda o, n: o * n,
                          divide=lambda o, n: o / n, max=np.maximum, min=np.minimum).items():
        assert np.array_equal(oracle.gradient_noise(g, old, "DEFAULT", blend, seed=7), fn(old, new).astype(np.float32)), blend


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


@pytest.mark.gpu
def test_gpu_seed_offsets(dev):
    capi, ctx = dev
    G = load_golden("noise_128x64_variants")
    for s, ref in zip(G["seeds"], G["seed_offsets"]):
        assert np.array_equal(capi.noise_seed_offset(int(s)), ref), s


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_gpu_noise_modes_golden(dev, mode):
    capi, ctx = dev
    G = load_golden("noise_128x64_modes")
    grid = capi.make_grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    out = ctx.gradient_noise(grid, G["img"], capi.noise_params(mode, seed=int(G["seed"]), warp=float(G["warp"])))
    assert np.array_equal(out, G["out_" + mode])


@pytest.mark.gpu
def test_gpu_noise_variants_golden(dev):
    capi, ctx = dev
    G = load_golden("noise_128x64_variants")
    grid = capi.make_grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    for key, mode, blend, sel, prm in _variants(G):
        out = ctx.gradient_noise(grid, G["img"], capi.noise_params(mode, blend, **prm), sel=sel)
        assert np.array_equal(out, G[key]), key


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_gpu_noise_vs_oracle(dev, oracle, mode):
    """other sizes (non-power-of-two included: kernel and oracle share the exact st), blends, parameters"""
    capi, ctx = dev
    rng = np.random.default_rng(5)
    for (W, H), blend, prm in (((300, 150), "replace", dict(seed=77, warp=1.1, scale=3.0, octaves=6)),
                               ((512, 256), "multiply", dict(seed=-3, lacunarity=1.9, persistence=0.6, octaves=11))):
        old = rng.normal(size=(H, W)).astype(np.float32)
        sel = rng.uniform(0, 1, (H, W)).astype(np.float32)
        ref = oracle.gradient_noise(oracle.grid(W, H), old, mode, blend, sel=sel, **prm)
        out = ctx.gradient_noise(capi.make_grid(W, H), old, capi.noise_params(mode, blend, **prm), sel=sel)
        assert np.array_equal(out, ref), (W, H, blend)


@pytest.mark.gpu
def test_gpu_ocean_generate_terrain(dev, oracle):
    """device-side terrain for an ocean handle == host-generated terrain uploaded: same Jacobi result, bands included"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    grid = capi.make_grid(W, H)
    prm = capi.noise_params("DEFAULT", seed=21, scale=1.5, min=-6.0, max=4.0)
    terrain = ctx.gradient_noise(grid, np.zeros((H, W), np.float32), prm)
    assert 0.05 < (terrain > 0).mean() < 0.95
    divw = synth.divw(W, H)
    a = capi.Ocean(ctx, grid)
    a.set_terrain(terrain); a.set_divw(divw); a.jacobi(64)
    pa = a.get(("pressure",))["pressure"]
    full = np.zeros_like(pa)
    for r in range(2):
        b = capi.Ocean(ctx, grid, band_rank=r, band_count=2, ghost=8)
        b.generate_terrain(prm); b.set_divw(divw)
        b.jacobi_sweeps(8)                    # within the ghost depth: no exchange needed
        full += b.get(("pressure",))["pressure"]
        b.close()
    c = capi.Ocean(ctx, grid)
    c.generate_terrain(prm); c.set_divw(divw); c.jacobi(64)
    assert np.array_equal(c.get(("pressure",))["pressure"], pa)
    d = capi.Ocean(ctx, grid)
    d.set_terrain(terrain); d.set_divw(divw); d.jacobi_sweeps(8)
    assert np.array_equal(full, d.get(("pressure",))["pressure"])


@pytest.mark.gpu
def test_gpu_noise_argument_errors(dev):
    """error behaviour of the boundary: bad enums / ranges are DMG_EINVAL with a message, nothing is written"""
    capi, ctx = dev
    grid = capi.make_grid(64, 32)
    f = np.zeros((32, 64), np.float32)
    for bad in (dict(mode=7), dict(mode=-1), dict(blend=9), dict(octaves=65), dict(octaves=-1)):
        p = capi.noise_params()
        for k, v in bad.items():
            setattr(p, k, v)
        with pytest.raises(capi.DmgError) as e:
            ctx.gradient_noise(grid, f, p)
        assert e.value.code == capi.EINVAL and "invalid argument" in str(e.value)
    oc = capi.Ocean(ctx, grid)
    with pytest.raises(capi.DmgError):
        oc.generate_terrain(capi.noise_params("DEFAULT", "add"))          # nothing to blend with
    oc.close()
    # octaves = 0 divides 0 by 0 in the reference's shader (fc /= total_amplitude): NaN there, NaN here
    out = ctx.gradient_noise(grid, f, capi.noise_params(octaves=0))
    assert np.isnan(out).all()


@pytest.mark.gpu
def test_gpu_noise_2048x1024_vs_oracle(dev, oracle):
    """2 M texels, default parameters (8 octaves), the three heaviest modes"""
    capi, ctx = dev
    W, H = 2048, 1024
    zero = np.zeros((H, W), np.float32)
    for mode in ("DEFAULT", "SWISS", "Plateaus"):
        ref = oracle.gradient_noise(oracle.grid(W, H), zero, mode, seed=12, warp=0.4)
        out = ctx.gradient_noise(capi.make_grid(W, H), zero, capi.noise_params(mode, seed=12, warp=0.4))
        assert np.array_equal(out, ref), mode
