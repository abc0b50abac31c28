"""SURVEY.md section 8(f) "next" rows: ThermalErosion and Morphological (erode / dilate / gradient).
CPU: the oracle restatement reproduces the llvmpipe goldens bit for bit.  GPU: the HIP kernels through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden, relerr


def test_oracle_thermal_golden(oracle):
    for n in ("full", "region"):
        G = load_golden("thermal_128x64_" + n)
        g = oracle.grid(128, 64, tuple(G["coords"]), float(G["circ"]))
        assert (G["out"] != G["h"]).sum() > 500                     # the case actually erodes
        assert np.array_equal(oracle.thermal_erosion(g, G["h"]), G["out"])


def test_oracle_morphological_golden(oracle):
    G = load_golden("morph_128x64_full")
    g = oracle.grid(128, 64)
    assert np.array_equal(oracle.morph_rlist(1.0), G["rlist_r1"]) and np.array_equal(oracle.morph_rlist(5.5), G["rlist_r5"])
    for key, (r, op) in dict(erode_r1=(1.0, "min"), dilate_r1=(1.0, "max"), erode_r5=(5.5, "min"), dilate_r5=(5.5, "max")).items():
        assert np.array_equal(oracle.morphological(g, G["field"], r, op), G[key]), key
    assert np.array_equal(oracle.morphological_gradient(g, G["field"], 3.0), G["gradient_r3"])


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", ["full", "region"])
def test_thermal_erosion(dev, oracle, n):
    capi, ctx = dev
    G = load_golden("thermal_128x64_" + n)
    grid = capi.make_grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    out = ctx.thermal_erosion(grid, G["h"])
    assert np.array_equal(out, G["out"])        # atan as Mesa expands it: the talus threshold falls where the reference's does


@pytest.mark.gpu
def test_morphological(dev, oracle):
    capi, ctx = dev
    G = load_golden("morph_128x64_full")
    grid = capi.make_grid(128, 64)
    for key, (r, op) in dict(erode_r1=(1.0, "min"), dilate_r1=(1.0, "max"), erode_r5=(5.5, "min"), dilate_r5=(5.5, "max")).items():
        assert np.array_equal(ctx.morphological(grid, G["field"], r, op), G[key]), key      # ring offsets come from the host
    assert np.array_equal(ctx.morphological_gradient(grid, G["field"], 3.0), G["gradient_r3"])
    from demiurge_amd import synth
    W, H = 1024, 512
    f = synth.terrain(W, H)
    assert np.array_equal(ctx.morphological(capi.make_grid(W, H), f, 4.0, "max"), oracle.morphological(oracle.grid(W, H), f, 4.0, "max"))
