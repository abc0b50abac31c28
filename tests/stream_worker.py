"""Worker for tests/test_gpu_jacobi.py: the streaming Jacobi kernel, forced for every 8-sweep launch through
DMG_TB_SHAPE (read once per process, hence a child process), against the ORACLE and the reference's llvmpipe golden
on coast-heavy inputs -- isolated land cells, random starting pressure, band splits with split launches.
Exit code 0 = everything bit-identical.

usage: python tests/stream_worker.py <case> [...]      cases: random, pepper, golden, bands, split, open, twin
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))

f32 = np.float32
# STREAM_WORKER_FORM=3 (round 6): every case in the tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE) against the oracle's form 3 --
# bit for bit, like the exact cases: launches of 8 sweeps in the fused form, the last 1..7 sweeps of a count in the exact form
FORM = int(os.environ.get("STREAM_WORKER_FORM", "0"))
BELT = (f32(-1.0), f32(1.2), f32(-np.pi), f32(np.pi))


def ogrid(oracle, grid):
    return oracle.grid(grid.W, grid.H, tuple(grid.coords), grid.circumference)


def solve(capi, ctx, grid, img, b, iters, p0=None, bands=1, ghost=1, block=8, split=False, names=None):
    """one band, or `bands` bands on this GPU with slab copies between blocks of `block` sweeps; split: every block but
    the last goes through _begin / exchange / _end (the two halves of the block's last launch around the exchange)"""
    from test_gpu_jacobi import exchange
    prm = capi.default_ocean_params()
    prm.jacobi_kernel = capi.JACOBI_BLOCKED
    if FORM == 3:
        prm.quirks |= capi.QUIRK_JACOBI_TOLERANCE
    oc = [capi.Ocean(ctx, grid, params=prm, band_rank=r, band_count=bands, ghost=ghost) for r in range(bands)]
    for o in oc:
        o.set_terrain(img)
        o.set_divw(b)
        if p0 is None:
            o.zero_pressure()
        else:
            o.set_pressure(p0)
    done = 0
    if bands > 1 and split:
        exchange(capi, ctx, oc, ghost)
    while done < iters:
        n = min(block if bands > 1 else iters, iters - done)
        if bands > 1 and split and done + n < iters:
            for o in oc:
                o.jacobi_sweeps_begin(n)
            exchange(capi, ctx, oc, ghost)
            for o in oc:
                o.jacobi_sweeps_end()
        else:
            if bands > 1 and not split:
                exchange(capi, ctx, oc, ghost)
            for o in oc:
                o.jacobi_sweeps(n)
            if bands > 1 and split and done + n < iters:
                exchange(capi, ctx, oc, ghost)
        done += n
    out = {"pressure": np.zeros((grid.H, grid.W), f32)}
    for o in oc:
        o.get(want=("pressure",), out=out)
        if names is not None:
            names.add(o.kernel_name())
        o.close()
    return out["pressure"]


def ref_solve(oracle, g, img, b, iters, p0=None, rows=True, nthreads=0):
    """the oracle's solve in the form under test"""
    if FORM != 3:
        return oracle.ocean_jacobi(g, img, b, iters, p0=p0, rows=rows, nthreads=nthreads)
    n8 = iters // 8 * 8
    p = oracle.ocean_jacobi_form(g, img, b, n8, 3, p0=p0, nthreads=nthreads) if n8 else p0
    return oracle.ocean_jacobi(g, img, b, iters - n8, p0=p, rows=True, nthreads=nthreads) if iters > n8 else p


def same(tag, got, ref):
    bad = got.view(np.uint32) != ref.view(np.uint32)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("FAIL", tag, int(bad.sum()), "cells differ; rows", ys.min(), "..", ys.max(), "cols", xs.min(), "..", xs.max(),
              [(int(y), int(x), float(got[y, x]), float(ref[y, x])) for y, x in zip(ys[:4], xs[:4])], flush=True)
        return False
    print("ok  ", tag, flush=True)
    return True


def case_random(capi, ctx, oracle, names):
    """small random geometries: widths from 256 up, heights from a few rows to several units, land from none to nearly
    all, salt-and-pepper and blocky, pole and belt canvases, random start, sweep counts of one to five launches"""
    rng = np.random.default_rng(20261004)
    ok = True
    for case in range(20):
        W = int(rng.choice([256, 264, 320, 488, 512, 1000]))
        H = int(rng.choice([16, 47, 48, 64, 97, 130, 300]))
        iters = int(rng.choice([8, 16, 24, 13, 40, 29]))
        coords = BELT if case % 3 == 2 else capi.FULL_GLOBE
        grid = capi.make_grid(W, H, coords)
        land = rng.random()
        img = np.where(rng.random((H, W)) < land * 0.5, 1.0, -1.0).astype(f32)
        if case % 4 == 0:
            img = np.repeat(np.repeat(np.where(rng.random((H // 2 + 1, W // 64 + 1)) < land, 1.0, -1.0), 2, 0), 64, 1)[:H, :W].astype(f32)
        b = rng.standard_normal((H, W)).astype(f32)
        p0 = rng.standard_normal((H, W)).astype(f32)
        ref = ref_solve(oracle, ogrid(oracle, grid), img, b, iters, p0=p0, rows=True)
        ok &= same("random %d: %dx%d x%d %s land %.2f" % (case, W, H, iters, "belt" if coords is BELT else "globe", land),
                   solve(capi, ctx, grid, img, b, iters, p0=p0, names=names), ref)
    return ok


def case_pepper(capi, ctx, oracle, names):
    """2048x1024, 12 % isolated land cells on top of the survey continents, random start: every land-neighbour count
    (0, 1, 2 per axis, land centres) in nearly every row; 24 sweeps = three streaming launches"""
    from demiurge_amd import synth
    rng = np.random.default_rng(7)
    W, H = 2048, 1024
    ok = True
    for coords, tag in ((capi.FULL_GLOBE, "globe"), (BELT, "belt")):
        grid = capi.make_grid(W, H, coords)
        img = synth.terrain(W, H).copy()
        img[rng.random((H, W)) < 0.12] = 1.0
        img[rng.random((H, W)) < 0.05] = -1.0            # ... and lakes inside the continents
        b = rng.standard_normal((H, W)).astype(f32)
        p0 = (rng.standard_normal((H, W)) * 3).astype(f32)
        ref = ref_solve(oracle, ogrid(oracle, grid), img, b, 24, p0=p0, rows=True)
        ok &= same("pepper 2048x1024 " + tag, solve(capi, ctx, grid, img, b, 24, p0=p0, names=names), ref)
    return ok


def case_golden(capi, ctx, oracle, names):
    """the reference's own GLSL output after its full 5000 sweeps (llvmpipe, 256x128), through 625 streaming launches"""
    from conftest import load_golden
    G = load_golden("jacobi_256x128_full")
    grid = capi.make_grid(256, 128, tuple(G["coords"]), float(G["circ"]))
    ok = True
    for iters, key in ((8, None), (104, None), (5000, "p5000")):
        got = solve(capi, ctx, grid, G["img"], G["divw"], iters, names=names)
        if key and FORM == 3:
            # (VERDICT r5 item 2's bound: half of north_star's 1e-5 against the reference's own pixels -- and not the same bits)
            err = float(np.max(np.abs(got.astype(np.float64) - G[key])) / np.max(np.abs(G[key])))
            print("tolerance rendering vs llvmpipe golden x%d: %.3g" % (iters, err), flush=True)
            ok &= err <= 5e-6 and not np.array_equal(got, G[key])
        elif key:
            ok &= same("golden 256x128 x%d vs llvmpipe" % iters, got, G[key])
        ok &= same("golden 256x128 x%d vs oracle" % iters, got, ref_solve(oracle, ogrid(oracle, grid), G["img"], G["divw"], iters, rows=True))
    return ok


def case_bands(capi, ctx, oracle, names):
    """band splits with ghost 8 and 16, whole launches between exchanges"""
    rng = np.random.default_rng(11)
    ok = True
    for W, H, bands, ghost, iters in ((512, 256, 2, 8, 40), (1024, 512, 4, 16, 48), (320, 200, 3, 8, 24)):
        grid = capi.make_grid(W, H)
        img = np.where(rng.random((H, W)) < 0.2, 1.0, -1.0).astype(f32)
        b = rng.standard_normal((H, W)).astype(f32)
        p0 = rng.standard_normal((H, W)).astype(f32)
        ref = ref_solve(oracle, ogrid(oracle, grid), img, b, iters, p0=p0, rows=True)
        ok &= same("bands %dx%d %d bands ghost %d" % (W, H, bands, ghost),
                   solve(capi, ctx, grid, img, b, iters, p0=p0, bands=bands, ghost=ghost, block=ghost, names=names), ref)
    return ok


def case_split(capi, ctx, oracle, names):
    """the two halves of a block's last launch around the exchange (_begin / _end), small and at a size above 4096
    tiles (16384 wide, two bands), where the unit heights are chosen per launch geometry"""
    from demiurge_amd import synth
    rng = np.random.default_rng(13)
    ok = True
    # (ghost 128 at 16384 wide: the send zones alone hold more short units than the chip has wave slots -- the interior's twin
    # units are a round of their own; round 4's first cut of them was not, and failed on exactly this geometry)
    for W, H, bands, ghost, iters in ((512, 256, 2, 8, 40), (1024, 768, 3, 16, 64), (16384, 2048, 2, 32, 96), (16384, 2048, 2, 8, 24), (16384, 2048, 2, 128, 256)):
        grid = capi.make_grid(W, H)
        if W < 16384:
            img = np.where(rng.random((H, W)) < 0.2, 1.0, -1.0).astype(f32)
        else:
            img = synth.terrain(W, H)
        b = rng.standard_normal((H, W)).astype(f32)
        p0 = rng.standard_normal((H, W)).astype(f32)
        ref = ref_solve(oracle, ogrid(oracle, grid), img, b, iters, p0=p0, rows=True, nthreads=16)
        ok &= same("split %dx%d %d bands ghost %d" % (W, H, bands, ghost),
                   solve(capi, ctx, grid, img, b, iters, p0=p0, bands=bands, ghost=ghost, block=ghost, split=True, names=names), ref)
    return ok


def case_open(capi, ctx, oracle, names):
    """open-water units (round 4: units whose whole window is plain open water run a program of their own, st_open_unit):
    grids without land (every interior unit), the survey continents (open and generic units side by side, open units next to
    coast units), odd heights (units of odd row counts: the tail steps), a belt (no pole rows), random start -- against the
    oracle.  DMG_ST_ROWS in the environment picks the unit height."""
    from demiurge_amd import synth
    rng = np.random.default_rng(41)
    ok = True
    for W, H, iters, terrain, coords in ((2048, 512, 24, "ocean", capi.FULL_GLOBE), (1000, 333, 40, "ocean", BELT), (4096, 1024, 24, "survey", capi.FULL_GLOBE),
                                         (2048, 777, 16, "survey", BELT), (512, 301, 29, "ocean", capi.FULL_GLOBE)):
        grid = capi.make_grid(W, H, coords)
        img = synth.terrain(W, H) if terrain == "survey" else -np.ones((H, W), f32)
        b = rng.standard_normal((H, W)).astype(f32)
        p0 = rng.standard_normal((H, W)).astype(f32)
        ref = ref_solve(oracle, ogrid(oracle, grid), img, b, iters, p0=p0, rows=True, nthreads=16)
        ok &= same("open units %dx%d x%d %s %s" % (W, H, iters, terrain, "belt" if coords is BELT else "globe"),
                   solve(capi, ctx, grid, img, b, iters, p0=p0, names=names), ref)
    return ok


def case_twin(capi, ctx, oracle, names):
    """twin units (round 4: pairs of units that start on a common seam and hand each other their first row of every level
    while their pipelines fill, k_jacobi_stream_twin): seams in open water, on coast rows and next to rows that are land
    everywhere, pairs with an empty or a one-row side (short stretches of water between wide land), poles and belts, odd
    heights, random start, one to five launches -- against the oracle.  Launches of up to 1200 rows take the twin cut
    whenever the simulated dispatch prefers it; DMG_ST_TWIN=2 in the environment forces it."""
    from demiurge_amd import synth
    rng = np.random.default_rng(43)
    ok = True
    cases = [(2048, 512, 24, "ocean", capi.FULL_GLOBE), (1000, 333, 40, "ocean", BELT), (4096, 1024, 24, "survey", capi.FULL_GLOBE),
             (2048, 777, 16, "survey", BELT), (512, 301, 29, "ocean", capi.FULL_GLOBE), (1024, 200, 16, "stripes", capi.FULL_GLOBE),
             (512, 64, 8, "stripes", BELT), (16384, 1024, 16, "survey", capi.FULL_GLOBE), (768, 1100, 13, "pepper", capi.FULL_GLOBE)]
    for W, H, iters, terrain, coords in cases:
        grid = capi.make_grid(W, H, coords)
        if terrain == "survey":
            img = synth.terrain(W, H).copy()
        elif terrain == "stripes":
            # bands of land of 1 .. 40 rows between bands of water of 1 .. 30 rows, different from strip to strip
            img = -np.ones((H, W), f32)
            for x0 in range(0, W, 240):
                y = int(rng.integers(0, 12))
                while y < H:
                    y += int(rng.integers(1, 31))
                    h = int(rng.integers(1, 41))
                    img[y:y + h, x0:x0 + 240] = 1.0
                    y += h
        elif terrain == "pepper":
            img = np.where(rng.random((H, W)) < 0.15, 1.0, -1.0).astype(f32)
        else:
            img = -np.ones((H, W), f32)
        b = rng.standard_normal((H, W)).astype(f32)
        p0 = rng.standard_normal((H, W)).astype(f32)
        ref = ref_solve(oracle, ogrid(oracle, grid), img, b, iters, p0=p0, rows=True, nthreads=16)
        ok &= same("twin units %dx%d x%d %s %s" % (W, H, iters, terrain, "belt" if coords is BELT else "globe"),
                   solve(capi, ctx, grid, img, b, iters, p0=p0, names=names), ref)
    return ok


class _OracleMemo:
    """The oracle's Jacobi solve, remembered on disk for the length of a test session: half a dozen tests run these same cases under
    different kernel switches, each in a child process of its own, and the CPU solve (seconds to tens of seconds per case on the
    box's 16 cores) is most of their time.  The key is a hash of everything the solve reads."""

    def __init__(self, oracle):
        import hashlib, tempfile
        self.o, self.hashlib = oracle, hashlib
        self.dir = Path(tempfile.gettempdir()) / "dmg_oracle_memo"
        self.dir.mkdir(exist_ok=True)

    def __getattr__(self, name):
        return getattr(self.o, name)

    def ocean_jacobi(self, g, img, divw, iters, p0=None, rows=False, nthreads=0):
        h = self.hashlib.sha1()
        h.update(repr((g.W, g.H, tuple(g.coords), g.circumference, iters, bool(rows))).encode())
        for a in (img, divw, p0):
            h.update(b"-" if a is None else np.ascontiguousarray(a, dtype=np.float32).tobytes())
        f = self.dir / (h.hexdigest() + ".npy")
        if f.exists():
            try:
                return np.load(f)
            except Exception:
                pass
        out = self.o.ocean_jacobi(g, img, divw, iters, p0=p0, rows=rows, nthreads=nthreads)
        tmp = self.dir / ("%s.%d.tmp.npy" % (h.hexdigest(), os.getpid()))
        np.save(tmp, out)
        os.replace(tmp, f)
        return out


def main():
    import oracle
    oracle = _OracleMemo(oracle)
    from demiurge_amd import capi
    ctx = capi.Context(0)
    names = set()
    ok = True
    for c in sys.argv[1:]:
        ok &= {"random": case_random, "pepper": case_pepper, "golden": case_golden, "bands": case_bands, "split": case_split, "open": case_open, "twin": case_twin}[c](capi, ctx, oracle, names)
    print("kernels:", sorted(names), "DMG_TB_SHAPE =", os.environ.get("DMG_TB_SHAPE"), flush=True)
    ctx.close()
    if not ok:
        sys.exit(1)
    print("bit-identical")


if __name__ == "__main__":
    main()
