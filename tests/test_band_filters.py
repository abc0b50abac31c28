"""Latitude-band forms of the generic filters on a fake cluster: n bands of one field on GPU 0, ghost rows
refreshed with dmg_fields_exchange / device copies of the dmg_*_halo slabs.  Every band-split result must equal the
whole-grid call of include/demiurge_hip.h bit for bit (full globe: wrap + both poles; and a regional extent: clamps)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REGION = (-0.6, 0.9, -1.0, 1.4)


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def field_case(W, H, seed=3):
    from demiurge_amd import synth
    return (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, seed).reshape(H, W) * 8).astype(np.float32)


def grids(capi, W, H):
    return [capi.make_grid(W, H), capi.make_grid(W, H, REGION)]


@pytest.mark.parametrize("bands,radius", [(2, 3.0), (3, 10.0), (4, 1.0)])
def test_band_blur_is_bitwise(dev, bands, radius):
    capi, ctx = dev
    W, H = 512, 256
    f0 = field_case(W, H)
    for grid in grids(capi, W, H):
        ref = ctx.blur(grid, f0, radius)
        rl = capi.blur_schedule(radius)
        ghost = max(capi.blur_halo(r, True) for r in rl)
        assert ghost <= H // bands
        fs = [capi.Field(ctx, grid, r, bands, ghost, data=f0) for r in range(bands)]
        for r in rl:                                               # Blur::step: (0, r) then (r, 0), BlurMenu.cpp:97-117
            for vertical in (True, False):
                capi.fields_exchange(fs, capi.blur_halo(r, vertical))
                for f in fs:
                    f.blur_pass(r, vertical)
        got = np.zeros((H, W), np.float32)
        for f in fs:
            f.get(got)
            f.close()
        assert np.array_equal(got, ref), (bands, radius)


def test_band_blur_is_bitwise_through_the_streamed_pass():
    """The same with the (0, r) pass streamed down the columns (k_blur13_ytile), which grids of this size do not get by themselves:
    DMG_BLUR_YT_ROWS forces it (read once per process: a child runs the test above), so its loads stay inside a band's ghost rows
    and its row records follow a band's first row."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    for rows in ("8", "24"):
        res = subprocess.run([sys.executable, "-m", "pytest", str(Path(__file__).resolve()), "-q", "-m", "gpu", "-k", "test_band_blur_is_bitwise and not streamed",
                              "-p", "no:cacheprovider"], cwd=str(root), env=dict(os.environ, DMG_TEST_HOOKS="1", DMG_BLUR_YT_ROWS=rows),
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and "3 passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("bands", [2, 3])
def test_band_erosion_pass_is_bitwise(dev, bands):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    h0 = field_case(W, H) / np.float32(50)
    flow = np.abs(synth.lcg_uniform(W * H, 9).reshape(H, W)).astype(np.float32) * 3
    up = (np.maximum(h0, 0) * 0.5).astype(np.float32)
    for grid in grids(capi, W, H):
        ref = h0
        for _ in range(3):
            ref = ctx.erosion_pass(grid, ref, flow, up, 1.5, 1.2)
        hs = [capi.Field(ctx, grid, r, bands, 1, data=h0) for r in range(bands)]
        fl = [capi.Field(ctx, grid, r, bands, 1, data=flow) for r in range(bands)]
        us = [capi.Field(ctx, grid, r, bands, 1, data=up) for r in range(bands)]
        for _ in range(3):
            capi.fields_exchange(hs, 1)
            for h, f, u in zip(hs, fl, us):
                h.erosion_pass(f, u, 1.5, 1.2)
        got = np.zeros((H, W), np.float32)
        for x in hs:
            x.get(got)
        for x in hs + fl + us:
            x.close()
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("bands", [2, 4])
def test_band_thermal_erosion_is_bitwise(dev, bands):
    capi, ctx = dev
    W, H = 512, 256
    h0 = field_case(W, H, seed=5)
    for grid in grids(capi, W, H):
        ref = ctx.thermal_erosion(grid, h0)                       # ten rounds, ThermalErosion.cpp:26
        hs = [capi.Field(ctx, grid, r, bands, 2, data=h0) for r in range(bands)]
        ls = [capi.Field(ctx, grid, r, bands, 2) for r in range(bands)]
        for _ in range(10):
            capi.fields_exchange(hs, 2)
            for h, l in zip(hs, ls):
                h.thermal_round(l)
        got = np.zeros((H, W), np.float32)
        for x in hs:
            x.get(got)
        for x in hs + ls:
            x.close()
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("bands,ghost", [(2, 1), (3, 4)])
def test_band_climate_is_bitwise(dev, bands, ghost):
    """Temperature::run's step on bands: `ghost` steps between halo refreshes, the computed region shrinking a row a step"""
    capi, ctx = dev
    W, H, steps = 512, 256, 9
    terrain = field_case(W, H)
    T0 = (15 - 40 * np.abs(np.linspace(-1, 1, H))[:, None] + np.zeros((1, W))).astype(np.float32)
    for grid in grids(capi, W, H):
        k = capi.Climate(ctx, grid, T0, terrain)
        k.steps(3, steps)
        ref = k.get()
        k.close()
        ks = [capi.Climate(ctx, grid, T0, terrain, r, bands, ghost) for r in range(bands)]
        done = 0
        while done < steps:
            n = min(ghost, steps - done)
            for i, kk in enumerate(ks):                            # halo refresh: device copies of the dmg_climate_halo slabs
                if i + 1 < bands:
                    s_lo, r_lo, nb = kk.halo(1, ghost)             # my north slab <-> the next band's south slab
                    s_hi, r_hi, _ = ks[i + 1].halo(0, ghost)
                    ctx.copy_device(r_hi, s_lo, nb)
                    ctx.copy_device(r_lo, s_hi, nb)
            for kk in ks:
                kk.steps(3 + done, n)
            done += n
        got = np.zeros((H, W), np.float32)
        for kk in ks:
            kk.get(got)
            kk.close()
        assert np.array_equal(got, ref), (bands, ghost)
