"""Bitwise check of the 8-sweep Jacobi launch against single-sweep launches at a given size (GPU).

usage: python tools/stream_check.py [W H [sweeps [survey|ocean [globe|belt]]]]
(environment: DMG_TB_SHAPE=stream forces the streaming kernel; belt = latitudes -1 .. 1.2 rad, both edges
 clamp instead of reflecting through a pole)
"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
terrain = synth.terrain(W, H) if (len(sys.argv) <= 4 or sys.argv[4] == "survey") else -np.ones((H, W), np.float32)
belt = len(sys.argv) > 5 and sys.argv[5] == "belt"
grid = capi.make_grid(W, H, (-1.0, 1.2, -np.pi, np.pi)) if belt else capi.make_grid(W, H)
res = {}
ctx = capi.Context()
if True:
    for name, kern in (("single", capi.JACOBI_SINGLE), ("auto", capi.JACOBI_AUTO)):
        prm = capi.default_ocean_params()
        prm.jacobi_kernel = kern
        o = capi.Ocean(ctx, grid, prm)
        o.set_terrain(terrain)
        o.set_divw(synth.divw(W, H))
        t0 = time.perf_counter()
        o.jacobi(sweeps, zero_first=True)
        res[name] = o.get(want=("pressure",))["pressure"]
        print(name, o.kernel_name(), "%.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        o.close()
a, b = res["single"].view(np.uint32), res["auto"].view(np.uint32)
bad = a != b
print("cells that differ:", int(bad.sum()), "of", a.size)
if bad.any():
    ys, xs = np.nonzero(bad)
    print("rows", ys.min(), "..", ys.max(), " cols", xs.min(), "..", xs.max())
    print("first:", [(int(y), int(x), float(res['single'][y, x]), float(res['auto'][y, x])) for y, x in zip(ys[:5], xs[:5])])
    sys.exit(1)
print("bit-identical")
