#!/usr/bin/env python3
"""Per-step clocks inside the units of ONE launch of band r of an n-way split (experiments build: make EXTRA=-DST_EXPERIMENTS,
DMG_LIB=<that library>; DMG_ST_STEPPROF).  Twin units: slot 62 = the wave's entry, 0..21 = top of iteration k, 22..42 = the even
step's rows have arrived, 43..62 = the even step's levels are done, 63 = the unit's end (100 MHz clock).
    DMG_LIB=demiurge_amd/libdemiurge_hip_exp.so python tools/band_steps.py <n> <r>"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
n, r = int(sys.argv[1]), int(sys.argv[2])
out = ROOT / "gpurun_out" / ("band_steps_%d_%d.csv" % (n, r))
os.environ["DMG_ST_STEPPROF"] = str(out)
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from demiurge_amd import capi, synth
W, H = 16384, 8192
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
oc = capi.Ocean(ctx, capi.make_grid(W, H), band_rank=r, band_count=n, ghost=32)
oc.set_terrain(synth.terrain(W, H)); oc.set_divw(synth.divw(W, H))
for _ in range(12):
    oc.jacobi_sweeps(32)
ctx.sync()
oc.close()
a = np.loadtxt(out, delimiter=",", dtype=np.uint64).astype(np.int64)
a = a[a[:, 63] > 0]
entry, end = a[:, 62], a[:, 63]
base = entry.min()
top, arrived = a[:, 0:22], a[:, 22:42]
its = (top > 0).sum(1)
mode = np.bincount(its).argmax()
s = its == mode
print("band %d of %d: %d units recorded, %d with %d iterations (the most common); launch span %.1f us" % (r, n, len(a), s.sum(), mode, (end.max() - base) / 100.0))
f = lambda x: " ".join("%5.2f" % v for v in x)
q = lambda x: " ".join("%.1f" % v for v in np.percentile(x / 100.0, [0, 10, 50, 90, 100]))
print("entry after the launch's first wave (min p10 p50 p90 max):", q(entry - base))
print("entry -> top of iteration 0 (prologue)                    :", q(top[s, 0] - entry[s]))
print("unit: entry -> end                                        :", q(end[s] - entry[s]))
d = np.diff(np.concatenate([top[s, :mode], end[s, None]], 1), axis=1) / 100.0
print("iteration k (two steps), median us :", f(np.median(d, 0)))
print("iteration k, p90                   :", f(np.percentile(d, 90, 0)))
w = (arrived[s, :mode] - top[s, :mode]) / 100.0
print("of which: wait for the even step's rows, median:", f(np.median(w, 0)))
