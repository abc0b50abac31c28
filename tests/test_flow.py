"""Flow accumulation (FlowFilter) and the stream-power erosion loop (cpufilter::run).

CPU part: the oracle's restatement of the GLSL passes against the llvmpipe goldens, and structural properties of the
oracle's flow solve.  GPU part: the HIP pipeline against the oracle, BIT FOR BIT (codes, masks, lake graph, fp32 sums in
the reference's order, double pow rounded once), the erosion loop likewise, and conservation / determinism at BASELINE
config-2 size (4096x2048)."""
import numpy as np
import pytest

from conftest import load_golden, relerr


def test_oracle_flow_passes_golden(oracle):
    G = load_golden("flow_passes_128x64_full")
    H, W = G["blurred"].shape
    g = oracle.grid(W, H, tuple(G["coords"]), float(G["circ"]))
    codes = oracle.flow_direction(g, G["blurred"])
    # the direction goes through GLSL atan(): a stochastic-rounding threshold may flip on an ulp; <= 0.1 % of land
    assert np.array_equal(codes, G["codes"])                       # hash() and mix() in Mesa's forms: every code
    assert np.array_equal(oracle.flow_mask(g, G["blurred"], G["codes"]), G["mask"])
    lid = np.where(G["label"] < 0, np.float32(-1), (G["label"] + 1073741824).astype(np.int32).view(np.float32))
    assert np.array_equal(oracle.lake_adjacency(g, lid.astype(np.float32)), G["adjacency"])
    S = load_golden("flow_passes_128x64_sel")
    c2 = oracle.flow_direction(g, S["blurred"], S["sel"])
    assert np.array_equal(c2, S["codes"])
    assert np.array_equal(oracle.flow_mask(g, S["blurred"], S["codes"], S["sel"]), S["mask"])


def area_rows(W, H, circ=42000.0):
    y = (np.arange(H, dtype=np.float32) / np.float32(H))
    geoy = (y * (np.float32(np.pi / 2) - np.float32(-np.pi / 2)) + np.float32(-np.pi / 2)).astype(np.float32)
    pwx = np.float32(circ * (np.float32(np.pi) - np.float32(-np.pi)) / (2 * np.pi) / W)
    pwy = np.float32(circ * (np.float32(np.pi / 2) - np.float32(-np.pi / 2)) / (2 * np.pi) / H)
    return (np.float64(pwy * pwx) * np.cos(geoy).astype(np.float32) * 1e-5).astype(np.float32)


def check_flow_properties(r, W, H):
    """every drained cell's flow = its own area + the flows of the cells draining into it (+ a connected lake);
    here checked in aggregate: the river mouths together carry the area of all drained cells"""
    flow, mask, label = r["flow"], r["mask"], r["label"]
    drained = flow > 0
    assert drained.sum() > 0 and (flow[~drained] == -1).all()
    own = np.repeat(area_rows(W, H)[:, None], W, axis=1)
    m = mask.astype(np.int64)
    mouths = (mask >= 0) & ((m & 16) != 0) & ((m & 512) != 0)
    total_out = flow[mouths & drained].astype(np.float64).sum()
    total_area = own[drained].astype(np.float64).sum()
    assert abs(total_out - total_area) / total_area < 1e-4
    # labels: a sink labels itself; every labelled cell is land
    sinks = (mask >= 0) & ((m & 16) != 0)
    idx = np.arange(W * H).reshape(H, W)
    assert (label[sinks] == idx[sinks]).all() and (mask[label >= 0] >= 0).all()


def test_oracle_flow_properties(oracle):
    from demiurge_amd import synth
    W, H = 128, 64
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    r = oracle.flow_accumulate(oracle.grid(W, H), h, 0.5, 1.0)
    check_flow_properties(r, W, H)


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,exponent", [(128, 64, 1.0), (256, 128, 0.5), (100, 50, 1.0), (250, 125, 0.5)])
def test_flow_matches_oracle(dev, oracle, W, H, exponent):
    capi, ctx = dev
    from demiurge_amd import synth
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    ref = oracle.flow_accumulate(oracle.grid(W, H), h, 0.5, exponent)
    got = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, exponent)
    assert np.array_equal(got, ref["flow"]), (int((got != ref["flow"]).sum()), relerr(got, ref["flow"]))


@pytest.mark.gpu
def test_flow_without_a_river_mouth_and_after(dev, oracle):
    """no cell at or below the sea: no river mouth, no lake is ever reached, the flow is -1 everywhere (FlowFilter.cpp:533-604).  The
    second solve of such a grid looks for a mouth first and returns (dmg_ctx::flow_hint_no_mouths); a terrain WITH mouths after it
    must not be taken for one without.  All three against the oracle."""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 192, 96
    g, og = capi.make_grid(W, H), oracle.grid(W, H)
    dry = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 3).reshape(H, W) * 8).astype(np.float32)
    dry = dry - dry.min() + np.float32(5.0)
    wet = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 3).reshape(H, W) * 8).astype(np.float32)
    ref_dry = oracle.flow_accumulate(og, dry, 0.5, 1.0)["flow"]
    assert (ref_dry == -1).all()                         # the case is what it says
    for h, ref in ((dry, ref_dry), (dry, ref_dry), (dry + np.float32(1.0), None), (wet, None), (dry, ref_dry), (wet, None)):
        if ref is None:
            ref = oracle.flow_accumulate(og, h, 0.5, 1.0)["flow"]
        got = ctx.flow_accumulate(g, h, 0.5, 1.0)
        assert np.array_equal(got, ref), int((got != ref).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("exponent", [0.5, 0.37, 1.0])
def test_flow_matches_oracle_512(dev, oracle, exponent):
    """131 k cells, rougher terrain (more lakes), the erosion tool's exponent 0.5 and an odd one"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 256
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 7).reshape(H, W) * 20).astype(np.float32)
    ref = oracle.flow_accumulate(oracle.grid(W, H), h, 0.5, exponent)["flow"]
    got = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, exponent)
    assert np.array_equal(got, ref), (int((got != ref).sum()), relerr(got, ref))


@pytest.mark.gpu
def test_flow_mouth_filter_flag(dev, oracle):
    """`if (Nthbit(c.from,10)) continue;` (FlowFilter.cpp:552) is replicated by default and can be switched off"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 1024, 512                                           # cell indices with bit 512 set exist from W = 1024 on
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 3).reshape(H, W) * 20).astype(np.float32)
    g, grid = oracle.grid(W, H), capi.make_grid(W, H)
    lit = oracle.flow_accumulate_flags(g, h, 0.5, 0.5)
    fixed = oracle.flow_accumulate_flags(g, h, 0.5, 0.5, fix_mouth_filter=1)
    assert not np.array_equal(lit, fixed)                      # the literal line changes the lake graph on this case
    assert np.array_equal(ctx.flow_accumulate(grid, h, 0.5, 0.5), lit)
    assert np.array_equal(ctx.flow_accumulate(grid, h, 0.5, 0.5, lakes=2), fixed)


@pytest.mark.gpu
def test_flow_selection_mask(dev, oracle):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 128, 64
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    sel = np.ones((H, W), np.float32)
    sel[:, 40:60] = 0
    ref = oracle.flow_accumulate(oracle.grid(W, H), h, 0.5, 1.0, sel=sel)["flow"]
    got = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, 1.0, sel=sel)
    assert (got[:, 40:60] == -1).all()
    assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_flow_config2_conservation(dev):
    """BASELINE config 2 grid, 4096x2048: the river mouths carry the area of every drained cell"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 4096, 2048
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
    flow = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, 1.0)
    drained = flow > 0
    assert drained.sum() > 0.3 * W * H * 0.3
    own = np.repeat(area_rows(W, H)[:, None], W, axis=1)
    # a river mouth = a drained cell that no drained cell drains further: total outflow = sum over cells that are
    # local maxima of flow along their path; cheaper equivalent: sum of flow over cells minus sum over their parents
    # is not available without the codes, so use the global bound instead
    assert flow.max() <= own[drained].astype(np.float64).sum() * (1 + 1e-4)
    assert np.isfinite(flow).all() and (flow[~drained] == -1).all()


@pytest.mark.gpu
def test_erosion_run_matches_oracle(dev, oracle):
    """cpufilter::run, 4 rounds of { FlowFilter ; erosion pass } against the same loop built from oracle pieces"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, N, iters = 128, 64, 50, 4
    h0 = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    g = oracle.grid(W, H)
    up = (np.maximum(h0, 0) / np.float32(N)).astype(np.float32)
    hd = np.where(h0 <= 0, h0, h0 / np.float32(N)).astype(np.float32)
    for i in range(iters):
        flow = oracle.flow_accumulate(g, h0 if i == 0 else hd, 0.5, 0.5)["flow"]
        hd = oracle.erosion_pass(g, hd, flow, up, 1.0, 1.0)
    got = ctx.erosion_run(capi.make_grid(W, H), h0, N, iters, 0.5, 1.0, 1.0)
    assert np.array_equal(got, hd), (int((got != hd).sum()), relerr(got, hd))


@pytest.mark.gpu
def test_erosion_run_matches_oracle_512(dev, oracle):
    """the same loop at 512x256 with the slope exponent off 1 (pow in the erosion pass too), 3 rounds"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, N, iters = 512, 256, 50, 3
    h0 = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 11).reshape(H, W) * 12).astype(np.float32)
    g = oracle.grid(W, H)
    up = (np.maximum(h0, 0) / np.float32(N)).astype(np.float32)
    hd = np.where(h0 <= 0, h0, h0 / np.float32(N)).astype(np.float32)
    for i in range(iters):
        flow = oracle.flow_accumulate(g, h0 if i == 0 else hd, 0.5, 0.5)["flow"]
        hd = oracle.erosion_pass(g, hd, flow, up, 1.5, 1.2)
    got = ctx.erosion_run(capi.make_grid(W, H), h0, N, iters, 0.5, 1.2, 1.5)
    assert np.array_equal(got, hd), (int((got != hd).sum()), relerr(got, hd))


@pytest.mark.gpu
def test_erosion_config2_at_size(dev):
    """BASELINE config 2 (4096x2048 stream-power erosion): a few rounds of dmg_erosion_run at full size --
    deterministic (two runs agree bit for bit), finite, sea untouched, land mass only moved by uplift - erosion"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, N, iters = 4096, 2048, 50, 3
    h0 = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
    grid = capi.make_grid(W, H)
    a = ctx.erosion_run(grid, h0, N, iters, 0.5, 1.0, 1.0)
    b = ctx.erosion_run(grid, h0, N, iters, 0.5, 1.0, 1.0)
    assert np.array_equal(a, b)
    assert np.isfinite(a).all()
    sea = h0 <= 0
    assert np.array_equal(a[sea], h0[sea])                      # `if (h<=0) return;` (cpufilter.cpp:126)
    scaled = (h0[~sea] / np.float32(N)).astype(np.float32)
    # fc = terrain + min(hdiff, max(0, up - flow)) (:197): a round never adds more than hdiff <= tan(30 deg) * the
    # diagonal pixel distance (flow = -1 on cells no river mouth reaches makes `up - flow` exceed the uplift)
    diag = np.hypot(42000.0 / W, 42000.0 / (2 * H))
    assert (a[~sea] <= scaled + iters * np.tan(np.pi / 6) * diag * (1 + 1e-5)).all()
    assert (a[~sea] - scaled).max() > 0                         # something was uplifted
    # the flow solve behind it conserves area at this size (river mouths carry every drained cell)
    flow = ctx.flow_accumulate(grid, h0, 0.5, 1.0)
    drained = flow > 0
    own = np.repeat(area_rows(W, H)[:, None], W, axis=1)
    m = ctx.flow_passes(grid, h0, 0.5)[1].astype(np.int64)
    mouths = (m >= 0) & ((m & 16) != 0) & ((m & 512) != 0)
    total_out = flow[mouths & drained].astype(np.float64).sum()
    total_area = own[drained].astype(np.float64).sum()
    assert abs(total_out - total_area) / total_area < 1e-4


@pytest.mark.gpu
def test_erosion_handle_round_by_round(dev, oracle):
    """dmg_erosion (heights resident, stepped round by round as the editor shows them) = dmg_erosion_run; with the
    menu's lake toggle the FlowFilter of every round draws its lakes (cpufilter.cpp:95)"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H, N = 128, 64, 50
    h0 = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    grid = capi.make_grid(W, H)
    e = capi.Erosion(ctx, grid, h0, N, 0.5, 1.0, 1.0)
    e.rounds(1)
    first = e.get()
    e.rounds(3)
    got = e.get()
    e.close()
    assert np.array_equal(got, ctx.erosion_run(grid, h0, N, 4, 0.5, 1.0, 1.0))
    assert np.array_equal(first, ctx.erosion_run(grid, h0, N, 1, 0.5, 1.0, 1.0))
    # lakes: the same loop built from the one-shot pieces
    g = oracle.grid(W, H)
    up = (np.maximum(h0, 0) / np.float32(N)).astype(np.float32)
    hd = np.where(h0 <= 0, h0, h0 / np.float32(N)).astype(np.float32)
    for i in range(2):
        flow = ctx.flow_accumulate(grid, h0 if i == 0 else hd, 0.5, 0.5, lakes=1)
        hd = ctx.erosion_pass(grid, hd, flow, up, 1.0, 1.0)
    e = capi.Erosion(ctx, grid, h0, N, 0.5, 1.0, 1.0, lakes=1)
    e.rounds(2)
    assert np.array_equal(e.get(), hd)
    e.close()


@pytest.mark.gpu
def test_flow_lake_drawing(dev, oracle):
    """FlowFilter lakeflag (the standalone Flow tool passes true, FlowFilter.cpp:24-27): lake surfaces are 0"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 256, 128
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 99).reshape(H, W) * 8).astype(np.float32)
    ref = oracle.flow_accumulate_lakes(oracle.grid(W, H), h, 0.5, 1.0)
    got = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, 1.0, lakes=1)
    assert (ref == 0).sum() > 50                          # the case has lakes
    assert np.array_equal(got, ref)


def _flow_child(code, env_extra, timeout=900):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_flow_suite_through_the_parallel_growth():
    """the growth of the lake graph runs as independent cascades on all host cores above 100 000 lakes
    (flow.hip, k_growth_*).  DMG_FLOW_GROWTH=parallel forces that path at every size: this file's oracle comparisons (flow,
    lakes, selection, the mouth-filter flag, the erosion loops) once more through it, in a child process (the variable is
    read once)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMG_FLOW_GROWTH="parallel")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_flow.py"), "-q", "-m", "gpu", "-x", "-k",
                        "matches_oracle or mouth_filter or selection or lake_drawing or round_by_round or conservation"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


_GROWTH_CHILD = """
import sys, time, hashlib
import numpy as np
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
ctx = capi.Context(0)
for W, H, seed, amp in ((4096, 2048, 5, 3.0), (16384, 8192, 11, 6.0)):
    h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, seed).reshape(H, W) * amp).astype(np.float32)
    for lakes in (0, 1):
        t0 = time.perf_counter()
        f = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, 0.5, lakes=lakes)
        print(W, H, lakes, hashlib.sha256(f.tobytes()).hexdigest(), '%.3f s' % (time.perf_counter() - t0), flush=True)
ctx.close()
"""


@pytest.mark.gpu
def test_flow_parallel_growth_equals_serial_at_size():
    """the same flow fields, bit for bit (SHA-256), from the one-thread growth and from the cascades on all cores, at
    BASELINE config 2's 4096x2048 and config 5's 16384x8192 (2.7 M lakes, 11 M lake pairs), with and without lake drawing"""
    outs = {}
    for mode in ("serial", "parallel"):
        r = _flow_child(_GROWTH_CHILD, {"DMG_FLOW_GROWTH": mode})
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = [l.split()[:4] for l in r.stdout.strip().splitlines()]
    assert len(outs["serial"]) == 4 and outs["serial"] == outs["parallel"], (outs["serial"], outs["parallel"])


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DMG_FLOW_WALK": "0"}, {"DMG_FLOW_TWO_PHASE": "0"}, {"DMG_FLOW_LOCAL_LIMIT": "12"}],
                         ids=["phase-B-dataflow", "one-pass", "one-pass-after-phase-A (a catchment deeper than the keys hold)"])
def test_flow_suite_through_the_other_sum_forms(env):
    """Round 6: the upstream sums are formed in two phases -- inside the catchments beside the host's growth, then along the reaches
    behind the connections (flow.hip, k_flow_catchment_sums / k_flow_walk).  The forms they replaced stay as fallbacks (a catchment
    deeper than the keys of phase A have bits for; more levels than the walk's keys hold): this file's oracle comparisons once more
    through each of them, in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_flow.py"), "-q", "-m", "gpu", "-x", "-k",
                        "matches_oracle or mouth_filter or selection or lake_drawing or round_by_round or conservation"],
                       cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_flow_sum_forms_equal_at_size():
    """the same flow fields, bit for bit (SHA-256), from the two-phase sums with the walk along the reaches (the default), with phase B
    as a dataflow, and from the one pass in depth order, at 4096x2048 and 16384x8192, with and without lake drawing"""
    outs = {}
    for name, env in (("walk", {}), ("dataflow", {"DMG_FLOW_WALK": "0"}), ("one pass", {"DMG_FLOW_TWO_PHASE": "0"})):
        r = _flow_child(_GROWTH_CHILD, env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = [l.split()[:4] for l in r.stdout.strip().splitlines()]
    assert len(outs["walk"]) == 4 and outs["walk"] == outs["dataflow"] == outs["one pass"], outs


_LAKES_CHILD = """
import sys, hashlib
import numpy as np
sys.path.insert(0, '.')
from demiurge_amd import capi, synth
ctx = capi.Context(0)
for name, W, H in (("noise", 2048, 1024), ("plateaus", 4096, 2048)):
    if name == "noise":          # a lake every few cells: a million catchments of a handful of cells
        h = (synth.lcg_uniform(W * H, 17).reshape(H, W) * 50).astype(np.float32)
    else:                        # a handful of flat values, as the tectonics render pass leaves them: every cell a lake of its own, 8.4 M of them
        h = np.floor(synth.terrain(W, H) * 6).astype(np.float32)
    for lakes in (0, 1):
        f = ctx.flow_accumulate(capi.make_grid(W, H), h, 0.5, 0.5, lakes=lakes)
        print(name, lakes, hashlib.sha256(f.tobytes()).hexdigest(), flush=True)
ctx.close()
"""


@pytest.mark.gpu
def test_flow_sum_forms_equal_on_many_small_lakes():
    """the two-phase sums give the smallest catchments a LANE each (k_flow_tiny_catchment_sums) and step aside for the one pass where the
    lakes' ids no longer fit phase A's keys (four million: the plate map the tectonics pass leaves is 8.4 M one-cell lakes at 4096x2048):
    both cases against the one-pass form, bit for bit"""
    outs = {}
    for name, env in (("two phases", {}), ("one pass", {"DMG_FLOW_TWO_PHASE": "0"})):
        r = _flow_child(_LAKES_CHILD, env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = r.stdout.strip().splitlines()
    assert len(outs["two phases"]) == 4 and outs["two phases"] == outs["one pass"], outs
