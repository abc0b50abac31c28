"""Pins the Tectonics restatement (oracle/tect_oracle.c + oracle.tectonics) against the reference's own GLSL and host
loop run on llvmpipe (tests/golden/tectonics_*.npz, made by tests/golden/make_goldens.py tectonics).  CPU only.

Every pass gathers with GL_NEAREST, so parity is counted in texels that differ, not in a norm."""
import numpy as np
import pytest

from conftest import load_golden


def _bad(a, b):
    return int((np.any(a != b, axis=-1) if a.ndim == 3 else (a != b)).sum())


def _run(oracle, G, trace=None):
    H, W = G["img"].shape
    g = oracle.grid(W, H, tuple(G["coords"]), float(G["circ"]))
    return oracle.tectonics(g, G["img"], int(G["steps"]), trace=trace)


@pytest.mark.parametrize("name", ["globe_64x32_s70", "globe_128x64_s10", "belt_128x64_s12", "region_128x64_s12"])
def test_tectonics_golden_bit_exact(oracle, name):
    """power-of-two grids: bit for bit, over the reference's full 70 steps on the smallest grid"""
    G = load_golden("tectonics_" + name)
    tr = {}
    terrain, plates = _run(oracle, G, tr)
    assert _bad(tr["fold"], G["fold1"]) == 0
    assert _bad(tr["spread"], G["spread1"]) == 0
    assert _bad(terrain, G["terrain"]) == 0
    assert plates[0].shape == G["plate0"].shape and plates[1].shape == G["plate1"].shape   # the swap size shuffle
    assert _bad(plates[0], G["plate0"]) == 0 and _bad(plates[1], G["plate1"]) == 0


def test_tectonics_golden_non_pow2(oracle):
    """100x50: llvmpipe interpolates st a few 1e-8 off (x+0.5)/W, which re-breaks exact distance ties in the
    spreading passes on a handful of texels; the bound is 0.5 % of texels after 5 steps."""
    G = load_golden("tectonics_globe_100x50_s5")
    tr = {}
    terrain, plates = _run(oracle, G, tr)
    assert _bad(tr["fold"], G["fold1"]) == 0
    assert _bad(tr["spread"], G["spread1"]) <= 5
    assert _bad(terrain, G["terrain"]) <= 0.005 * terrain.size
    for k in range(2):
        assert _bad(plates[k], G["plate%d" % k]) <= 0.005 * plates[k].shape[0] * plates[k].shape[1]
    # with st as llvmpipe's rasteriser interpolates it on each render target (tools/st_probe.py) the spreading passes and
    # both plate textures are the reference's texel for texel after the 5 steps; one texel of the rendered terrain is not
    oracle.set_raster_st(True)
    try:
        tr = {}
        terrain, plates = _run(oracle, G, tr)
    finally:
        oracle.set_raster_st(False)
    assert _bad(tr["fold"], G["fold1"]) == 0 and _bad(tr["spread"], G["spread1"]) == 0
    assert _bad(plates[0], G["plate0"]) == 0 and _bad(plates[1], G["plate1"]) == 0
    assert _bad(terrain, G["terrain"]) <= 1


def test_tectonics_libm_would_not_pin(oracle):
    """the transcendental variant matters: with libm asin/atan the same run is NOT bit-exact (documents why the
    contract is Mesa's polynomial forms)"""
    G = load_golden("tectonics_globe_128x64_s10")
    oracle.tect_set_math(0)
    try:
        terrain, plates = _run(oracle, G)
    finally:
        oracle.tect_set_math(2)
    assert _bad(plates[0], G["plate0"]) > 0


def test_glm_rotate_properties(oracle):
    """glm::rotate restatement: orthonormal, accumulates angle about x, transpose is the inverse"""
    m = np.eye(4, dtype=np.float32)
    for _ in range(70):
        m = oracle.glm_rotate(m, np.float32(0.01), np.array([-1, 0, 0], np.float32))
    r = m[:3, :3].astype(np.float64)
    assert np.abs(r @ r.T - np.eye(3)).max() < 1e-5
    assert abs(np.arctan2(r[2][1], r[1][1]) - 0.7) < 1e-4 or abs(np.arctan2(r[1][2], r[1][1]) - 0.7) < 1e-4
