"""Parity of the HIP Jacobi pressure solve (through the C ABI) against the oracle and the llvmpipe goldens.
Integer-free fp32 path; contract: the kernels perform the GLSL's operations in the GLSL's order with IEEE
division, so they match the oracle's row-table form BIT FOR BIT; against the reference's own GLSL output the
bound is BASELINE.json's 1e-5 relative (max|a-ref|/max|ref|)."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from demiurge_amd import capi
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def run_jacobi(capi, ctx, grid, img, divw, iters, p0=None, bands=1, ghost=1, block=1, kernel=0, quirks=0):
    """Single GPU.  bands > 1 emulates the latitude-band split on one device: one dmg_ocean per band, halo
    slabs moved with dmg_copy_device between sweep blocks of `block` (<= ghost) sweeps."""
    prm = capi.default_ocean_params()
    prm.jacobi_kernel = kernel
    prm.quirks |= quirks
    oc = [capi.Ocean(ctx, grid, params=prm, band_rank=r, band_count=bands, ghost=ghost) for r in range(bands)]
    for o in oc:
        o.set_terrain(img)
        o.set_divw(divw)
        if p0 is None:
            o.zero_pressure()
        else:
            o.set_pressure(p0)
    done = 0
    while done < iters:
        n = min(block, iters - done)
        if bands > 1:
            exchange(capi, ctx, oc, ghost)
        for o in oc:
            o.jacobi_sweeps(n)
        done += n
    out = {"pressure": np.zeros((grid.H, grid.W), np.float32)}
    for o in oc:
        o.get(want=("pressure",), out=out)
        o.close()
    return out["pressure"]


def exchange(capi, ctx, oc, rows):
    for r in range(len(oc) - 1):
        s_send, s_recv, nb = oc[r].halo(capi.F_PRESSURE, capi.SIDE_NORTH, rows)
        n_send, n_recv, nb2 = oc[r + 1].halo(capi.F_PRESSURE, capi.SIDE_SOUTH, rows)
        assert nb == nb2 and nb > 0
        ctx.copy_device(n_recv, s_send, nb)      # band r's top rows -> band r+1's south ghosts
        ctx.copy_device(s_recv, n_send, nb)      # band r+1's bottom rows -> band r's north ghosts


def ogrid(oracle, grid):
    return oracle.grid(grid.W, grid.H, tuple(grid.coords), grid.circumference)


def test_headline_golden_and_oracle(dev, oracle):
    capi, ctx = dev
    G = load_golden("jacobi_256x128_full")
    grid = capi.make_grid(256, 128, tuple(G["coords"]), float(G["circ"]))
    for iters, key in ((1, "p1"), (100, "p100"), (5000, "p5000")):
        p = run_jacobi(capi, ctx, grid, G["img"], G["divw"], iters)
        assert np.array_equal(p, G[key]), (iters, relerr(p, G[key]))                     # vs the reference's GLSL: bit for bit, all 5000 sweeps
        ref = oracle.ocean_jacobi(ogrid(oracle, grid), G["img"], G["divw"], iters, rows=True)
        assert np.array_equal(p, ref), (iters, relerr(p, ref))                             # vs the oracle: bitwise


@pytest.mark.parametrize("name", ["full", "belt", "region"])
def test_extents_golden_and_oracle(dev, oracle, name):
    capi, ctx = dev
    G = load_golden("jacobi_128x64_" + name)
    grid = capi.make_grid(128, 64, tuple(G["coords"]), float(G["circ"]))
    for iters, key in ((1, "p1"), (60, "p60")):
        p = run_jacobi(capi, ctx, grid, G["img"], G["divw"], iters, p0=G["p0"])
        assert np.array_equal(p, G[key])                                                   # the reference's GLSL, bit for bit
        ref = oracle.ocean_jacobi(ogrid(oracle, grid), G["img"], G["divw"], iters, p0=G["p0"], rows=True)
        assert np.array_equal(p, ref)


@pytest.mark.parametrize("W,H", [(100, 50), (36, 18), (520, 96), (4, 2), (1024, 512)])
def test_ragged_sizes_bitwise(dev, oracle, W, H):
    """widths that are not multiples of the vector width / tile, tiny grids, config-1 size"""
    capi, ctx = dev
    from demiurge_amd import synth
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    p = run_jacobi(capi, ctx, grid, img, b, 23)
    assert np.array_equal(p, oracle.ocean_jacobi(ogrid(oracle, grid), img, b, 23, rows=True))


def test_all_land_and_all_ocean(dev, oracle):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 128, 64
    grid = capi.make_grid(W, H)
    b = synth.divw(W, H)
    for img in (np.ones((H, W), np.float32), -np.ones((H, W), np.float32), np.zeros((H, W), np.float32)):
        p = run_jacobi(capi, ctx, grid, img, b, 9)
        assert np.array_equal(p, oracle.ocean_jacobi(ogrid(oracle, grid), img, b, 9, rows=True))
    assert not run_jacobi(capi, ctx, grid, np.ones((H, W), np.float32), b, 3).any()


@pytest.mark.parametrize("bands,ghost,block", [(2, 1, 1), (4, 4, 4), (8, 3, 2), (3, 5, 5)])
def test_band_split_is_bitwise(dev, bands, ghost, block):
    """latitude bands + ghost-row exchange reproduce the single-band result exactly"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 256, 128
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    one = run_jacobi(capi, ctx, grid, img, b, 37)
    split = run_jacobi(capi, ctx, grid, img, b, 37, bands=bands, ghost=ghost, block=block)
    assert np.array_equal(one, split)


def test_full_size_bitwise_vs_oracle(dev, oracle):
    """BASELINE config: 16384x8192, a few sweeps against the oracle bit for bit, plus band split at full size."""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 16384, 8192
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    p = run_jacobi(capi, ctx, grid, img, b, 6)
    ref = oracle.ocean_jacobi(ogrid(oracle, grid), img, b, 6, rows=True)
    assert np.array_equal(p, ref)
    del ref
    split = run_jacobi(capi, ctx, grid, img, b, 6, bands=8, ghost=3, block=3)
    assert np.array_equal(p, split)


def test_errors(dev):
    capi, ctx = dev
    with pytest.raises(capi.DmgError):
        capi.Ocean(ctx, capi.make_grid(101, 50))           # odd width with poles
    with pytest.raises(capi.DmgError):
        capi.Ocean(ctx, capi.make_grid(2, 1))
    o = capi.Ocean(ctx, capi.make_grid(64, 32))
    with pytest.raises(capi.DmgError) as e:
        o.jacobi(1)                                         # before set_terrain
    assert e.value.code == capi.ESTATE
    o.close()


def test_halo_slabs_alias_as_torch_tensors(dev):
    """the zero-copy path bench.py uses for RCCL: library-owned device slabs viewed as torch tensors"""
    import torch
    capi, ctx = dev
    from demiurge_amd import bands, synth
    W, H, G = 256, 128, 4
    grid = capi.make_grid(W, H)
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal((H, W)).astype(np.float32)
    img = -np.ones((H, W), np.float32)
    o = capi.Ocean(ctx, grid, band_rank=1, band_count=4, ghost=G)     # owns rows 32..63
    o.set_terrain(img)
    o.set_pressure(p0)
    ctx.sync()

    class _D:   # minimal stand-in for torch.distributed: only get_backend() is consulted here
        @staticmethod
        def get_backend():
            return "nccl"
    ex = bands.HaloExchanger(_D, 1, 4, torch.device("cuda", 0))
    for side, send_rows, recv_rows in ((capi.SIDE_SOUTH, slice(32, 36), slice(28, 32)), (capi.SIDE_NORTH, slice(60, 64), slice(64, 68))):
        s, r, nb = o.halo(capi.F_PRESSURE, side, G)
        assert nb == G * W * 4
        ts, tr = ex._tensor(s, nb), ex._tensor(r, nb)
        assert ts.is_cuda and ts.data_ptr() == s
        assert np.array_equal(ts.cpu().numpy().reshape(G, W), p0[send_rows])
        assert np.array_equal(tr.cpu().numpy().reshape(G, W), p0[recv_rows])
        tr.copy_(torch.full_like(tr, 7.0))                             # writes land in the library's buffer
    torch.cuda.synchronize()
    o.close()


# ---- temporally blocked kernel (8 sweeps per launch) == single-sweep kernel, bit for bit -----------------------
@pytest.mark.parametrize("W,H,iters", [(256, 128, 8), (256, 128, 37), (512, 64, 5), (1000, 500, 19), (264, 16, 24), (2048, 1024, 100)])
def test_blocked_equals_single(dev, oracle, W, H, iters):
    capi, ctx = dev
    from demiurge_amd import synth
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    rng = np.random.default_rng(11)
    p0 = rng.standard_normal((H, W)).astype(np.float32)
    single = run_jacobi(capi, ctx, grid, img, b, iters, p0=p0, block=iters, kernel=capi.JACOBI_SINGLE)
    blocked = run_jacobi(capi, ctx, grid, img, b, iters, p0=p0, block=iters, kernel=capi.JACOBI_BLOCKED)
    assert np.array_equal(single, blocked)
    assert np.array_equal(blocked, oracle.ocean_jacobi(ogrid(oracle, grid), img, b, iters, p0=p0, rows=True))


def test_blocked_belt_no_poles(dev, oracle):
    capi, ctx = dev
    G = load_golden("jacobi_128x64_belt")
    from demiurge_amd import synth
    W, H = 512, 96
    grid = capi.make_grid(W, H, tuple(G["coords"]), float(G["circ"]))
    img, b = synth.terrain(W, H), synth.divw(W, H)
    got = run_jacobi(capi, ctx, grid, img, b, 21, block=21, kernel=capi.JACOBI_BLOCKED)
    assert np.array_equal(got, oracle.ocean_jacobi(ogrid(oracle, grid), img, b, 21, rows=True))


@pytest.mark.parametrize("bands,ghost,block", [(2, 8, 8), (4, 16, 16), (3, 16, 11)])
def test_blocked_band_split_is_bitwise(dev, bands, ghost, block):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 512, 384
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    one = run_jacobi(capi, ctx, grid, img, b, 45, block=45, kernel=capi.JACOBI_SINGLE)
    split = run_jacobi(capi, ctx, grid, img, b, 45, bands=bands, ghost=ghost, block=block, kernel=capi.JACOBI_BLOCKED)
    assert np.array_equal(one, split)


@pytest.mark.parametrize("bands,ghost,W,H", [(2, 32, 512, 384), (3, 16, 512, 384), (4, 8, 768, 256), (8, 32, 2048, 1024)])
def test_overlapped_block_order_is_bitwise(dev, bands, ghost, W, H):
    """dmg_ocean_jacobi_sweeps_begin/_end: the exchange for the next block is done between the edge tiles and the
    interior tiles of a block's last launch (what bands.jacobi_solve does with RCCL running alongside).  Here the copies
    happen on the same stream right after _begin, i.e. while the interior tiles have not run yet: they must neither
    need the ghost rows being overwritten nor change the rows being sent."""
    capi, ctx = dev
    from demiurge_amd import synth
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    iters = ghost * 17 + 5          # the first twelve blocks of a geometry are launched whole (kernel/shape tuner), then split
    one = run_jacobi(capi, ctx, grid, img, b, iters, block=iters, kernel=capi.JACOBI_SINGLE)
    prm = capi.default_ocean_params()
    prm.jacobi_kernel = capi.JACOBI_BLOCKED
    oc = [capi.Ocean(ctx, grid, params=prm, band_rank=r, band_count=bands, ghost=ghost) for r in range(bands)]
    for o in oc:
        o.set_terrain(img); o.set_divw(b); o.zero_pressure()
    exchange(capi, ctx, oc, ghost)
    done = 0
    while done < iters:
        n = min(ghost, iters - done)
        if done + n < iters:
            for o in oc:
                o.jacobi_sweeps_begin(n)
            if done >= 13 * ghost:
                with pytest.raises(capi.DmgError):
                    oc[0].get(want=("pressure",))         # the pressure is half a launch apart: refused
            exchange(capi, ctx, oc, ghost)                # slabs of the NEW iterate, interior tiles still to come
            for o in oc:
                o.jacobi_sweeps_end()
        else:
            for o in oc:
                o.jacobi_sweeps(n)
        done += n
    out = {"pressure": np.zeros((H, W), np.float32)}
    for o in oc:
        o.get(want=("pressure",), out=out)
        o.close()
    assert np.array_equal(out["pressure"], one)


def test_blocked_full_size_bitwise(dev, oracle):
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 16384, 8192
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    # five 8-sweep launches of the streaming kernel (the default above 4096 tiles) and a 4-sweep launch, against the oracle
    blocked = run_jacobi(capi, ctx, grid, img, b, 44, block=44, kernel=capi.JACOBI_BLOCKED)
    ref = oracle.ocean_jacobi(ogrid(oracle, grid), img, b, 44, rows=True, nthreads=16)
    assert np.array_equal(blocked, ref)


def test_tolerance_rendering_full_size(dev):
    """the pressure solve's tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE: fused row coefficients, three
    fused multiply-adds per cell instead of three divisions) against the exact rendering at the headline size, 16384x8192, after
    the reference's full 5000 sweeps from zero pressure: within north_star's 1e-5 of the field's maximum -- and not the same
    bits, or the switch would be doing nothing.  (The oracle's form 3 says 2.5e-6 at 2048x1024 and 3.6e-6 at 4096x2048.)"""
    capi, ctx = dev
    from demiurge_amd import synth
    W, H = 16384, 8192
    grid = capi.make_grid(W, H)
    img, b = synth.terrain(W, H), synth.divw(W, H)
    exact = run_jacobi(capi, ctx, grid, img, b, 5000, block=5000, kernel=capi.JACOBI_BLOCKED)
    tol = run_jacobi(capi, ctx, grid, img, b, 5000, block=5000, kernel=capi.JACOBI_BLOCKED, quirks=capi.QUIRK_JACOBI_TOLERANCE)
    m = float(np.max(np.abs(exact)))
    err = 0.0
    for y in range(0, H, 1024):                              # (in slices: a float64 copy of the grid is 1 GiB)
        err = max(err, float(np.max(np.abs(tol[y:y + 1024].astype(np.float64) - exact[y:y + 1024]))))
    print("tolerance rendering vs exact, 16384x8192 x5000: max|d| / max|ref| = %.3g" % (err / m))
    assert 0 < err / m <= 1e-5, err / m
    assert np.array_equal(tol[img > 0], exact[img > 0])      # land: exactly +0.0 in both


def test_blocked_randomised_geometries(dev, oracle):
    """many small random cases: widths that are multiples of 8 from 256 up, heights down to 2 tile rows or less, land
    fractions from none to nearly all, sweep counts that are not multiples of 8, pole and belt canvases, band splits"""
    capi, ctx = dev
    rng = np.random.default_rng(20261003)
    belt = (np.float32(-1.0), np.float32(1.2), np.float32(-np.pi), np.float32(np.pi))
    for case in range(24):
        W = int(rng.choice([256, 264, 320, 488, 512, 1000]))
        H = int(rng.choice([2, 6, 16, 47, 48, 49, 64, 97, 130]))
        iters = int(rng.integers(1, 21))
        coords = belt if case % 3 == 2 else capi.FULL_GLOBE
        grid = capi.make_grid(W, H, coords)
        land = rng.random()
        img = np.where(rng.random((H, W)) < land * 0.5, 1.0, -1.0).astype(np.float32)
        if case % 4 == 0:                                  # blocky continents: whole wave-rows of land
            img = np.repeat(np.repeat(np.where(rng.random((H // 2 + 1, W // 64 + 1)) < land, 1.0, -1.0), 2, 0), 64, 1)[:H, :W].astype(np.float32)
        b = rng.standard_normal((H, W)).astype(np.float32)
        p0 = rng.standard_normal((H, W)).astype(np.float32)
        ref = oracle.ocean_jacobi(ogrid(oracle, grid), img, b, iters, p0=p0, rows=True)
        got = run_jacobi(capi, ctx, grid, img, b, iters, p0=p0, block=iters, kernel=capi.JACOBI_BLOCKED)
        assert np.array_equal(got, ref), (case, W, H, iters)
        if H >= 48 and case % 2 == 0:
            bands = 2 if H < 97 else 3
            ghost = min(16, H // bands)
            got = run_jacobi(capi, ctx, grid, img, b, iters, p0=p0, bands=bands, ghost=ghost, block=min(ghost, iters),
                             kernel=capi.JACOBI_BLOCKED)
            assert np.array_equal(got, ref), ("bands", case, W, H, iters)


def test_llvmpipe_raster_quirk(dev, oracle):
    """DMG_QUIRK_LLVMPIPE_RASTER: the per-fragment kernel with llvmpipe's rasteriser `st` and float-weight taps gives the
    software GL path's pixels on grids that are not powers of two -- the 96x64 golden (300 sweeps) and the reference's
    default 1000x500 (vs the oracle in the same mode, itself pinned to llvmpipe there by default_1000x500_step.npz) --
    and the bits of the default kernels on a power-of-two grid"""
    capi, ctx = dev
    from demiurge_amd import synth
    prm = capi.default_ocean_params()
    prm.quirks |= capi.QUIRK_LLVMPIPE_RASTER
    G = load_golden("jacobi_96x64_full")
    o = capi.Ocean(ctx, capi.make_grid(96, 64, tuple(G["coords"]), float(G["circ"])), params=prm)
    o.set_terrain(G["img"]); o.set_divw(G["divw"]); o.jacobi(300)
    assert np.array_equal(o.get(want=("pressure",))["pressure"], G["p300"])
    o.close()
    W, H = 1000, 500
    img, b = synth.terrain(W, H), synth.divw(W, H)
    o = capi.Ocean(ctx, capi.make_grid(W, H), params=prm)
    o.set_terrain(img); o.set_divw(b); o.jacobi(50)
    got = o.get(want=("pressure",))["pressure"]
    o.close()
    oracle.set_linear_taps(True); oracle.set_raster_st(True)
    try:
        want = oracle.ocean_jacobi(oracle.grid(W, H), img, b, 50)
    finally:
        oracle.set_raster_st(False); oracle.set_linear_taps(False)
    assert np.array_equal(got, want)
    W, H = 256, 128
    img, b = synth.terrain(W, H), synth.divw(W, H)
    outs = []
    for q in (prm, capi.default_ocean_params()):
        o = capi.Ocean(ctx, capi.make_grid(W, H), params=q)
        o.set_terrain(img); o.set_divw(b); o.jacobi(40)
        outs.append(o.get(want=("pressure",))["pressure"]); o.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("canvas", ["globe", "belt"])
@pytest.mark.parametrize("form", ["stream"])
def test_streaming_forms_forced_are_bitwise(form, canvas):
    """The streaming kernel (round 5 retired its narrow and persistent forms: no launch of a BASELINE size ever chose them), forced for all
    8-sweep launches of a 2048x1024 solve through DMG_TB_SHAPE, against single-sweep launches, bit for bit, on the globe
    (rows beyond a pole are the pole rows shifted by W/2) and on a latitude belt (rows beyond an edge are plain mirrors).
    The environment is read once per process, hence the child process (tools/stream_check.py does the comparison)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMG_TB_SHAPE=form)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_check.py"), "2048", "1024", "64", "survey", canvas], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    kern = {"stream": "k_jacobi_stream "}[form]
    assert kern.strip() in r.stdout and "bit-identical" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("rows", ["24", "100", "232"])
def test_streaming_unit_heights_are_bitwise(rows):
    """The unit form with its shortest, a middling and its tallest units (232 rows: the window rows' classes fill all four
    bytes of the class register, the coefficient table all of its LDS), against single-sweep launches, bit for bit."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMG_TB_SHAPE="stream", DMG_ST_ROWS=rows)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_check.py"), "2048", "1024", "40", "survey", "globe"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bit-identical" in r.stdout and "k_jacobi_stream" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _stream_worker(env_extra, cases, timeout=900):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stream_worker.py")] + cases, cwd=root, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stream"])
def test_streaming_forms_forced_vs_oracle(form):
    """the streaming kernel, forced for all 8-sweep launches, against the ORACLE on
    coast-heavy inputs with random starting pressure (tests/stream_worker.py: 20 random geometries, salt-and-pepper land
    at 2048x1024 on globe and belt, band splits with ghost 8 and 16), and against the reference's own llvmpipe output
    after its full 5000 sweeps (256x128 golden, 625 streaming launches)."""
    # (DMG_ST_TWIN=0: the plain units; the twin units that small launches take by default have their own test below)
    out = _stream_worker({"DMG_TB_SHAPE": form, "DMG_ST_TWIN": "0"}, ["random", "pepper", "golden", "bands"])
    kern = {"stream": "k_jacobi_stream"}[form]
    assert "'%s'" % kern in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"DMG_JACOBI_DIV2": "0"}, {"DMG_DIV2_FORCE_BAD": "7"}, {"DMG_ST_TWIN": "1"}],
                         ids=["forced", "three-op-division", "every-7th-row-without-pair", "where-the-simulation-prefers-them"])
def test_streaming_twin_units_vs_oracle(env):
    """Round 4: launches of about one round (the band of an 8-way or 4-way split, every small grid) are cut into TWIN units --
    two waves of one workgroup that start on the two sides of a seam row and stream away from it, handing each other their
    first row of every level through LDS while their pipelines fill: no halo rows between them (k_jacobi_stream_twin).
    Forced for every launch of up to 2400 rows (DMG_ST_TWIN=2) and as the default choice: seams in open water, on coast rows,
    next to rows that are land everywhere, pairs with an empty or a one-row side, poles and belts, odd heights, 16384 x 1024
    (the 8-way band itself), the 20 random geometries, salt-and-pepper coasts, the 5000-sweep golden against llvmpipe, band
    splits with whole and with split launches (the interior half of a split launch is twin units, its send zones plain
    ones) -- against the oracle, in both division forms and beside rows without an exact divisor pair."""
    e = dict({"DMG_TB_SHAPE": "stream", "DMG_ST_TWIN": "2"}, **env)
    # (every case with twin units forced; the division forms on the cases that vary the terrain; the default choice on the band cases)
    cases = (["twin", "random", "golden", "bands", "split", "pepper"] if not env else
             ["twin", "bands", "split"] if "DMG_ST_TWIN" in env else ["twin", "random", "golden"])
    out = _stream_worker(e, cases, timeout=1500)
    assert "'k_jacobi_stream_twin'" in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("rendering", ["exact", "tolerance"])
def test_deep_form_vs_oracle(rendering):
    """Round 6: the deep form of the streaming kernel (k_jacobi_deep: two waves per unit, the front wave's level 8 handed to the
    back wave's level 0 through LDS, SIXTEEN sweeps per trip through HBM) widened from round 5's open-water prototype to
    everything a unit handles -- land, coast, poles, band edges.  It is the default for the 16-sweep launches of the tolerance
    rendering (where half the bytes per sweep pay) and can be forced in the exact one (DMG_JACOBI_DEEP=1): 20 random geometries,
    salt-and-pepper coasts, band splits (ghost 16: the bands' launches are deep ones), open water, the 5000-sweep golden -- bit for
    bit against the oracle (the exact form, and form 3) and, in the exact rendering, against the reference's own llvmpipe pixels."""
    env = {"DMG_TB_SHAPE": "stream", "DMG_ST_TWIN": "0", "DMG_JACOBI_DEEP": "1"}
    if rendering == "tolerance":
        env["STREAM_WORKER_FORM"] = "3"
    out = _stream_worker(env, ["random", "pepper", "golden", "bands", "open", "split"], timeout=1500)
    assert "'k_jacobi_deep'" in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("twin", ["0", "2"], ids=["plain-units", "twin-units"])
def test_streaming_tolerance_rendering_vs_oracle_form3(twin):
    """Round 6: DMG_QUIRK_JACOBI_TOLERANCE -- the sweep with fused row coefficients, a = ah + al, b = bh + bl,
    ah + bh = 0.5 exactly -- is the oracle's form 3 (oracle/dmg_oracle.c, orc_ocean_jacobi_rows_form) BIT FOR BIT on the cases of
    the exact kernel: 20 random geometries, salt-and-pepper coasts, band splits with whole and split launches, open water,
    twin units; sweep counts that are no multiple of 8 end in the exact arithmetic on both sides.  Against the reference's own
    pixels (llvmpipe golden, 256x128, 5000 sweeps) it is within 5e-6 of the field's maximum."""
    cases = ["random", "pepper", "golden", "bands", "open"] + (["twin", "split"] if twin == "2" else [])
    # (DMG_JACOBI_DEEP=0: the 8-sweep kernel's fused arithmetic alone; the deep form that takes the 16-sweep launches by default: test_deep_form_vs_oracle)
    out = _stream_worker({"DMG_TB_SHAPE": "stream", "DMG_ST_TWIN": twin, "STREAM_WORKER_FORM": "3", "DMG_JACOBI_DEEP": "0"}, cases, timeout=1500)
    assert "tolerance rendering vs llvmpipe golden x5000" in out, out[-1500:]
    assert ("'k_jacobi_stream_twin'" if twin == "2" else "'k_jacobi_stream'") in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DMG_JACOBI_DIV2": "0"}, {"DMG_DIV2_FORCE_BAD": "7"}, {"DMG_DIV2_FORCE_BAD": "1"}],
                         ids=["three-op-division", "every-7th-row-without-pair", "no-row-with-pair"])
def test_streaming_division_forms_vs_oracle(env):
    """The default streaming kernel divides by the row constants in two operations (div2.hip).  Its other forms give the
    same bits: the three-operation division everywhere (DMG_JACOBI_DIV2=0), and rows that have no exact pair -- one in a
    thousand at 16384x8192, forced here for every 7th row and for every row -- which take the three-operation form row by
    row inside the two-operation kernel."""
    _stream_worker(dict(env, DMG_TB_SHAPE="stream", DMG_ST_TWIN="0"), ["random", "pepper", "golden"])


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"DMG_ST_ROWS": "24"}, {"DMG_ST_ROWS": "25"}, {"DMG_ST_ROWS": "100"}, {"DMG_ST_ROWS": "232"}, {"DMG_ST_OPEN_UNITS": "0"},
                                 {"DMG_DIV2_FORCE_BAD": "50"}, {"DMG_JACOBI_DIV2": "0"}],
                         ids=["auto-height", "24-rows", "25-rows", "100-rows", "232-rows", "without-open-units", "every-50th-row-without-pair", "three-op-division"])
def test_streaming_open_water_units_vs_oracle(env):
    """Round 4: units whose whole window is plain open water (the host flags them from the class table) can run a program of
    their own -- no classes, no code loads, no histories, no range tests (st_open_unit; DMG_ST_OPEN_UNITS=1: measured to gain
    nothing on a kernel that turned out to be bound by its bytes, so not the default).  Grids without land, the survey
    continents (open units next to coast units), odd heights, a belt canvas, random starting pressure, at unit heights from 24
    to 232 rows (even and odd: the tail steps), beside rows without an exact divisor pair (never inside an open unit) and in
    the three-operation division form: against the oracle.  With the open units switched off the same cases give the same bits."""
    out = _stream_worker(dict({"DMG_ST_OPEN_UNITS": "1", "DMG_ST_TWIN": "0"}, **dict(env, DMG_TB_SHAPE="stream", DMG_ST_DEBUG="1")), ["open"])
    assert "'k_jacobi_stream'" in out, out[-1500:]
    if env.get("DMG_ST_OPEN_UNITS") != "0":
        import re
        assert any(int(m) > 0 for m in re.findall(r"(\d+) of \d+ units are open-water units", out)), out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("flip", ["0", "1"])
def test_streaming_southward_units_vs_oracle(flip):
    """Round 4: every other segment of the streaming kernel's cut can stream SOUTHWARDS -- the same program on a mirrored view of
    the band (negative row stride, reversed row tables, mirrored poles, code bits swapped on arrival) -- so that neighbouring
    units start, or end, on the halo rows they share and one of them finds those rows in the L2.  With and without it: the
    random geometries (poles, belts, odd sizes, all-land and all-water rows), salt-and-pepper coasts, the 5000-sweep golden
    against llvmpipe, band splits and split launches, open-water grids at several unit heights -- against the oracle."""
    env = {"DMG_TB_SHAPE": "stream", "DMG_ST_FLIP": flip, "DMG_ST_DEBUG": "1", "DMG_ST_TWIN": "0"}       # (twin units: their own test)
    out = _stream_worker(env, ["random", "pepper", "golden", "bands", "split", "open"], timeout=1500)
    import re
    n = sum(int(m) for m in re.findall(r"(\d+) stream southwards", out))
    assert (n > 0) == (flip == "1"), out[-1500:]
    _stream_worker(dict(env, DMG_ST_ROWS="24"), ["open", "pepper"])
    _stream_worker(dict(env, DMG_DIV2_FORCE_BAD="7"), ["random"])


@pytest.mark.gpu
def test_twin_units_equal_plain_units_at_size():
    """Twin units against plain units where the oracle is too slow to referee: whole grids of 16384 columns and 777 to 4400 rows
    (survey, no land, salt-and-pepper, striped land), and the 16384x8192 solve in 2, 4 and 8 latitude bands on this GPU (peer
    copies, direct stores, RCCL loopback; both exchange orders, ghost 16 to 128) -- each run once with twin units forced and
    once with plain units, in child processes: the same SHA-256 every time, and one and the same field for one, two, four
    and eight bands (tools/twin_vs_plain.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "twin_vs_plain.py")], cwd=root, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0 and "ALL IDENTICAL" in r.stdout and "one field" in r.stdout, r.stdout[-3000:] + r.stderr[-1500:]
    assert r.stdout.count("k_jacobi_stream_twin") >= 12, r.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stream", "auto"])
def test_streaming_split_launches_vs_oracle(form):
    """the two halves of a split launch (_begin / exchange / _end) are halves of one cut.  Band splits
    through _begin/_end with the streaming kernel forced and with the default choice, including 16384-wide bands (above
    4096 tiles, where the unit heights are chosen per launch geometry), against the oracle."""
    _stream_worker({"DMG_TB_SHAPE": form} if form != "auto" else {}, ["split"], timeout=1500)


def test_div2_pairs_are_exact(dev):
    """dmg_div2_split: the (h, l) the streaming kernel divides with give the correctly rounded quotient -- here on a
    random sample of numerators in float64 arithmetic (the library itself tries all 2^23 significands on the device
    before it hands a pair out), for the Jacobi constants of a 16384x8192 globe including two whose standard pair fails
    for one significand, and a constant that has no pair at all."""
    capi, ctx = dev
    rng = np.random.default_rng(5)
    consts = [np.float32(3.7252903e-05), np.float32(float.fromhex("0x1.ba8f1ep+21")), np.float32(float.fromhex("0x1.dfc9b6p-16")),
              np.float32(1.0), np.float32(3.0), np.float32(0.1), np.float32(123456.789)]
    x = np.concatenate([rng.uniform(1, 2, 2_000_000), rng.uniform(1, 2, 1000) * 2.0 ** rng.integers(-60, 60, 1000)]).astype(np.float32)
    for c in consts:
        h, l = ctx.div2_split(c)
        # fma(x, h, x * l) evaluated exactly: x*l rounded to fp32, then x*h + u in float64 (exact: 48 + 24 bits) rounded once
        u = (x * l).astype(np.float32)
        q = (x.astype(np.float64) * np.float64(h) + u.astype(np.float64)).astype(np.float32)
        want = (x.astype(np.float64) / np.float64(c)).astype(np.float32)
        assert np.array_equal(q, want), (float(c), int((q != want).sum()))
    with pytest.raises(capi.DmgError):
        ctx.div2_split(np.float32(float.fromhex("0x1.a9407ep+16")))     # two significands pull l in opposite directions
    with pytest.raises(capi.DmgError):
        ctx.div2_split(np.float32(-2.0))
