"""The streaming pressure-solve kernel's unit lists, checked on the CPU (no GPU): `dmg_stream_units_probe` runs the host code
that cuts a launch's rows into units -- plain units, twin units, the halves of a split launch, the placement -- on a row-class
table and returns the list exactly as the kernel would read it.  What a list must be, whatever the terrain and the geometry:

  * every row of [lo, hi) that is not land everywhere is written by exactly one unit, and no unit reaches outside [lo, hi);
  * no unit is taller than the kernel's tables (232 rows);
  * a twin list is pairs: the same strip, one unit streaming northwards from the seam and one southwards to it;
  * the two halves of a split launch are halves of ONE cut: together they are the launch, the edge half is exactly the units
    that write rows of the send zones -- this is what the multi-GPU overlapped order stands on, and one GPU cannot run it
    between devices (round 4's first twin cut failed here for 8 bands with 128 ghost rows: found by a profile run, now a case).

These are properties of host code: the same library that runs the kernels, loaded without a device."""
import numpy as np
import pytest

from demiurge_amd import capi, synth

MAXROWS = 232


def classes(img, lo_row, hi_row):
    """row classes of the strips (240 output columns, 8 halo columns either side, columns wrap) for grid rows
    [lo_row, hi_row): 1 = land everywhere, 2 = a land cell in the row or next to it, 0 = open water"""
    H, W = img.shape
    land = img > 0
    n_strips = (W + 239) // 240
    out = np.zeros((n_strips, hi_row - lo_row), np.uint8)
    near = land | np.roll(land, 1, 0) | np.roll(land, -1, 0) | np.roll(land, 1, 1) | np.roll(land, -1, 1)
    for tx in range(n_strips):
        cols = np.arange(tx * 240 - 8, tx * 240 + 248) % W
        l, n = land[lo_row:hi_row][:, cols], near[lo_row:hi_row][:, cols]
        out[tx] = np.where(l.all(1), 1, np.where(n.any(1), 2, 0))
    return out


_survey = {}


def survey_classes(W, H, lo_row, hi_row):
    """the survey terrain's classes, computed once per grid size"""
    if (W, H) not in _survey:
        _survey[(W, H)] = classes(synth.terrain(W, H), 0, H)
    return np.ascontiguousarray(_survey[(W, H)][:, lo_row:hi_row])


def decode(units, L):
    """-> list of (strip, first row, end row, southward, twin, place holder) in the band's own row numbering"""
    res = []
    for tx, v0, v1, fl in units:
        south = bool(fl & 2)
        r0, r1 = (L - v1, L - v0) if south else (v0, v1)
        res.append((int(tx), int(r0), int(r1), south, bool(fl & 4), bool(fl & 8)))
    return res


def check_list(units, twin, rc, lo, hi, what):
    n_strips, L = rc.shape
    dec = decode(units, L)
    cover = np.zeros((n_strips, L), np.int32)
    for tx, r0, r1, south, tw, dummy in dec:
        if dummy:
            continue
        assert 0 <= tx < n_strips and lo <= r0 <= r1 <= hi, (what, tx, r0, r1)
        assert r1 - r0 <= MAXROWS, (what, tx, r0, r1)
        assert tw == bool(twin), what
        cover[tx, r0:r1] += 1
    assert cover.max() <= 1, (what, "rows written twice", np.argwhere(cover > 1)[:4])
    must = rc != 1
    must[:, :lo] = False
    must[:, hi:] = False
    missing = must & (cover == 0)
    assert not missing.any(), (what, "rows nobody writes", np.argwhere(missing)[:4])
    if twin:
        assert len(dec) % 2 == 0, what
        for a, b in zip(dec[0::2], dec[1::2]):
            assert a[5] == b[5], what
            if a[5]:
                continue
            n, s = (a, b) if not a[3] else (b, a)
            assert not n[3] and s[3] and n[0] == s[0] and n[1] == s[2], (what, "not a pair around a seam", a, b)
    return {(tx, r0, r1) for tx, r0, r1, _, _, dummy in dec if not dummy and r1 > r0}


def terrains(W, H, rng):
    yield "ocean", -np.ones((H, W), np.float32)
    yield "survey", synth.terrain(W, H)
    img = -np.ones((H, W), np.float32)                       # bands of land of 1..40 rows, different from strip to strip
    for x0 in range(0, W, 240):
        y = int(rng.integers(0, 12))
        while y < H:
            y += int(rng.integers(1, 31))
            h = int(rng.integers(1, 41))
            img[y:y + h, x0:x0 + 240] = 1.0
            y += h
    yield "stripes", img
    yield "pepper", np.where(rng.random((H, W)) < 0.15, 1.0, -1.0).astype(np.float32)


@pytest.mark.parametrize("W,H", [(2048, 1024), (16384, 1024), (1000, 333), (4096, 2048)])
def test_whole_launches_cover_their_rows_once(W, H):
    rng = np.random.default_rng(W + H)
    for name, img in terrains(W, H, rng):
        rc = classes(img, 0, H)
        for lo, hi in ((0, H), (8, H - 8), (0, H // 2), (H // 3, H)):
            units, twin = capi.stream_units_probe(W, H, rc, lo, hi)
            check_list(units, twin, rc, lo, hi, (name, W, H, lo, hi))
            again, twin2 = capi.stream_units_probe(W, H, rc, lo, hi)
            assert twin2 == twin and np.array_equal(units, again), (name, "not deterministic")


def test_the_full_size_launch_keeps_plain_units_and_bands_get_twin_units():
    W, H = 16384, 8192
    rc = survey_classes(W, H, 0, H)
    units, twin = capi.stream_units_probe(W, H, rc, 0, H)
    assert not twin and len(units) > 2048
    check_list(units, twin, rc, 0, H, "16384x8192")
    for bands in (2, 4, 8):
        rows, g = H // bands, 32
        rank = bands // 2
        gs, gn = g, (g if rank < bands - 1 else 0)
        y0 = rows * rank
        L = gs + rows + gn
        rcb = survey_classes(W, H, y0 - gs, y0 + rows + gn)
        lo, hi = 8, L - (8 if gn else 0)
        units, twin = capi.stream_units_probe(W, H, rcb, lo, hi, y0=y0, y1=y0 + rows, gs=gs, gn=gn)
        assert twin and len(units) <= 2048 + 16, (bands, len(units))
        check_list(units, twin, rcb, lo, hi, ("a band of", bands))


@pytest.mark.parametrize("W,bands,ghost", [(16384, 8, 32), (16384, 8, 128), (2048, 8, 8), (2048, 4, 16), (16384, 2, 64)])
def test_the_halves_of_a_split_launch_are_halves_of_one_cut(W, bands, ghost):
    """the geometry of dmg_ocean_jacobi_sweeps_begin (ocean.hip): after n = ghost sweeps since the halo refresh the block's last
    launch covers local rows [n, L - n) (from row 0 / up to row L at a pole), its send zones end at gs + ghost and start at
    L - gn - ghost -- for the first, a middle and the last band"""
    H = 8192 if W == 16384 else 1024
    rows = H // bands
    for rank in (0, bands // 2, bands - 1):
        gs, gn = (ghost if rank > 0 else 0), (ghost if rank < bands - 1 else 0)
        y0 = rows * rank
        L = gs + rows + gn
        rc = survey_classes(W, H, y0 - gs, y0 + rows + gn)
        n = ghost
        lo, hi = (n if gs else 0), L - (n if gn else 0)
        zone_lo = min(gs + ghost, hi) if gs else lo
        zone_hi = max(L - gn - ghost, lo) if gn else hi
        kw = dict(y0=y0, y1=y0 + rows, gs=gs, gn=gn)
        whole, tw0 = capi.stream_units_probe(W, H, rc, lo, hi, **kw)
        check_list(whole, tw0, rc, lo, hi, ("whole", W, bands, ghost, rank))
        edge, tw1 = capi.stream_units_probe(W, H, rc, lo, hi, part=1, zone_lo=zone_lo, zone_hi=zone_hi, **kw)
        inner, tw2 = capi.stream_units_probe(W, H, rc, lo, hi, part=2, zone_lo=zone_lo, zone_hi=zone_hi, **kw)
        # each half for itself writes no row twice; together they are the launch
        n_strips = rc.shape[0]
        cover = np.zeros((n_strips, L), np.int32)
        for part, (units, twin) in (("edge", (edge, tw1)), ("interior", (inner, tw2))):
            for tx, r0, r1, south, tw, dummy in decode(units, L):
                if dummy:
                    continue
                assert lo <= r0 <= r1 <= hi and r1 - r0 <= MAXROWS and tw == bool(twin), (part, tx, r0, r1)
                touches = r0 < zone_lo or r1 > zone_hi
                assert touches == (part == "edge"), (part, "unit on the wrong side", tx, r0, r1, zone_lo, zone_hi)
                cover[tx, r0:r1] += 1
        assert cover.max() <= 1, ("rows written by both halves", W, bands, ghost, rank, np.argwhere(cover > 1)[:4])
        must = rc != 1
        must[:, :lo] = False
        must[:, hi:] = False
        assert not (must & (cover == 0)).any(), ("rows neither half writes", W, bands, ghost, rank, np.argwhere(must & (cover == 0))[:4])


def test_probe_rejects_rows_outside_the_band():
    rc = np.zeros((9, 64), np.uint8)
    with pytest.raises(capi.DmgError):
        capi.stream_units_probe(2048, 64, rc, 0, 65)
    with pytest.raises(capi.DmgError):
        capi.stream_units_probe(2048, 64, rc, 10, 10)


def test_the_cut_does_not_depend_on_how_many_threads_judge_it():
    """Round 5: the candidate cuts and the stretches of a balanced cut are judged side by side on the host's cores (1.2 s -> 0.17 s
    for the full grid on the build container's 8).  The list is the one a single thread computes (DMG_ST_PLAN_THREADS=1, read once per
    process: child processes), for the full grid and for the middle band of an 8-way split (twin units)."""
    import hashlib
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = ("import sys, hashlib\nsys.path.insert(0, %r); sys.path.insert(0, %r)\nimport numpy as np\nfrom demiurge_amd import capi\n"
            "import test_stream_cut as t\nW, H = 8192, 4096\n"
            "for y0, y1, gs, gn in ((0, H, 0, 0), (1536, 2048, 32, 32)):\n"
            "    L = gs + (y1 - y0) + gn\n"
            "    r = np.ascontiguousarray(t.survey_classes(W, H, y0 - gs, y1 + gn))\n"
            "    u, tw = capi.stream_units_probe(W, H, r, 0, L, y0=y0, y1=y1, gs=gs, gn=gn)\n"
            "    print(len(u), tw, hashlib.sha256(u.tobytes()).hexdigest())\n" % (str(root), str(root / "tests")))
    outs = []
    for env in ({"DMG_ST_PLAN_THREADS": "1"}, {"DMG_ST_PLAN_THREADS": "3"}, {}):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.split())
    assert len(outs[0]) == 6 and outs[0] == outs[1] == outs[2], outs
