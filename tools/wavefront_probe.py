"""16 sweeps per trip through the grid WITHOUT a new kernel: the grid is cut into latitude chunks of C rows and every chunk
gets its two 8-sweep launches back to back -- A_c: rows [cC, (c+1)C) buffer 0 -> 1, B_c: rows [cC - 8, (c+1)C - 8) buffer 1 -> 0
(the last B up to H) -- so that B_c finds what A_c wrote, the right-hand sides and the codes still in the 256 MB Infinity Cache.
B_c overwrites buffer-0 rows no later A reads (A_{c+1} reads from row (c+1)C - 8 on).  Bit-identical to 16 plain sweeps.
usage: python tools/wavefront_probe.py [sweeps=320] [W=16384] [H=8192]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
H = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
ctx = capi.Context(0)
g = capi.make_grid(W, H)
img, b = synth.terrain(W, H), synth.divw(W, H)


def solve_plain(o, n):
    o.zero_pressure(); o.jacobi_sweeps(n)


def solve_wave(o, n, C):
    o.zero_pressure()
    assert n % 16 == 0
    cur = o.pressure_buffer()
    chunks = [(lo, min(lo + C, H)) for lo in range(0, H, C)]
    for _ in range(n // 16):
        for i, (lo, hi) in enumerate(chunks):
            o.jacobi_launch(lo, hi, 8, 0, 0, 0, cur)
            blo, bhi = max(lo - 8, 0), (H if i == len(chunks) - 1 else hi - 8)
            o.jacobi_launch(blo, bhi, 8, 0, 0, 0, cur ^ 1)
    o.set_pressure_buffer(cur, launches=2 * len(chunks) * (n // 16), sweeps=n)


def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best


for terrain_name, terr in (("survey", img), ("ocean", -np.ones((H, W), np.float32))):
    o = capi.Ocean(ctx, g)
    o.set_terrain(terr); o.set_divw(b)
    solve_plain(o, sweeps); ctx.sync()
    ref = o.get(want=("pressure",))["pressure"]
    ms = timed(lambda: solve_plain(o, sweeps))
    print("%s %dx%d plain: %.2f ms per %d sweeps = %.0f Mcell-steps/s" % (terrain_name, W, H, ms, sweeps, W * H * sweeps / ms / 1e3), flush=True)
    for C in (512, 1024, 2048, 4096):
        if C >= H:
            continue
        solve_wave(o, sweeps, C); ctx.sync()              # first use of every chunk geometry: its cut is planned here
        got = o.get(want=("pressure",))["pressure"]
        same = bool(np.array_equal(got, ref))
        ms = timed(lambda: solve_wave(o, sweeps, C))
        print("%s chunks of %4d rows: %.2f ms = %.0f Mcell-steps/s  bit-identical %s" % (terrain_name, C, ms, W * H * sweeps / ms / 1e3, same), flush=True)
    o.close()
