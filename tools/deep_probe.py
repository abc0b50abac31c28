"""(needs a library built with `make -C demiurge_amd/csrc EXTRA=-DDMG_EXPERIMENTS`)
The deep form's prototype (k_jacobi_deep_open: 16 sweeps per trip through HBM) on an all-ocean grid: bit-identity against 16 plain
sweeps on the rows it writes, and its time per 16 sweeps against two 8-sweep launches.  usage: python tools/deep_probe.py [W H]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
import ctypes
import os
_probe = ctypes.CDLL(os.environ.get("DMG_LIB") or str(Path(__file__).resolve().parents[1] / "demiurge_amd" / "libdemiurge_hip.so")).dmg_ocean_jacobi_deep_probe
_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
def deep(o, lo, hi, seg, src):
    capi.check(_probe(o.h, lo, hi, seg, src))
ctx = capi.Context(0)
g = capi.make_grid(W, H)
prm = capi.default_ocean_params()
if os.environ.get("DEEP_TOL") == "1":                 # round 6: the deep form in the tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE)
    prm.quirks |= capi.QUIRK_JACOBI_TOLERANCE
o = capi.Ocean(ctx, g, params=prm)
o.set_terrain(-np.ones((H, W), np.float32)); o.set_divw(synth.divw(W, H))
o.zero_pressure(); o.jacobi_sweeps(8)
p8 = o.get(want=("pressure",))["pressure"]
o.jacobi_sweeps(16)
ref = o.get(want=("pressure",))["pressure"]
lo, hi = 32, H - 32
import os
SEGS = [int(x) for x in os.environ.get("DEEP_SEGS", "64,120,200,232").split(",")]
for seg in SEGS:
    o.set_pressure(p8)
    cur = o.pressure_buffer()
    deep(o, lo, hi, seg, cur)
    o.set_pressure_buffer(cur ^ 1)
    got = o.get(want=("pressure",))["pressure"]
    bad = np.argwhere(got[lo:hi] != ref[lo:hi])
    print("seg %3d: rows [%d, %d) bit-identical to 16 plain sweeps: %s%s" % (seg, lo, hi, len(bad) == 0, "" if len(bad) == 0 else
          "  (%d cells differ; first at row %d col %d: %r vs %r; rows %s cols %s)" % (len(bad), bad[0][0] + lo, bad[0][1], got[lo + bad[0][0], bad[0][1]], ref[lo + bad[0][0], bad[0][1]],
                                                                                   sorted(set((bad[:, 0] + lo).tolist()))[:6], sorted(set(bad[:, 1].tolist()))[:8])), flush=True)
    o.set_pressure_buffer(cur)
# time: 20 x 16 sweeps
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.sync(); ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best
cur = o.pressure_buffer()
ms = timed(lambda: [o.jacobi_launch(0, H, 8, 0, 0, 0, cur ^ (i & 1)) for i in range(40)])
print("plain: 40 launches of 8 sweeps: %.2f ms = %.1f us per 16 sweeps -> %.0f Mcell-steps/s" % (ms, ms * 1e3 / 20, W * H * 320 / ms / 1e3), flush=True)
for seg in SEGS:
    ms = timed(lambda: [deep(o, lo, hi, seg, cur ^ (i & 1)) for i in range(20)])
    print("deep, units of %3d rows: 20 launches of 16 sweeps over rows [%d, %d): %.2f ms = %.1f us per launch -> %.0f Mcell-steps/s" %
          (seg, lo, hi, ms, ms * 1e3 / 20, W * (hi - lo) * 320 / ms / 1e3), flush=True)
