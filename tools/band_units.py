#!/usr/bin/env python3
"""Per-unit clocks of ONE launch of band r of an n-way split (DMG_ST_PROF), then tools/unit_times.py and tools/unit_simds.py on it.
    python tools/band_units.py <n> <r> [skip]     (skip: launches of more than 1024 units before the recorded one; default 9)"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
n, r = int(sys.argv[1]), int(sys.argv[2])
skip = sys.argv[3] if len(sys.argv) > 3 else "9"
out = ROOT / "gpurun_out" / ("band_units_%d_%d.csv" % (n, r))
os.environ["DMG_ST_PROF"] = str(out)
os.environ["DMG_ST_PROF_SKIP"] = skip
sys.path.insert(0, str(ROOT))
import torch
from demiurge_amd import capi, synth
W, H = 16384, 8192
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
oc = capi.Ocean(ctx, capi.make_grid(W, H), band_rank=r, band_count=n, ghost=32)
oc.set_terrain(synth.terrain(W, H)); oc.set_divw(synth.divw(W, H))
for _ in range(8):
    oc.jacobi_sweeps(32)
ctx.sync()
oc.close()
print("== band %d of %d, launch after %s warm ones" % (r, n, skip), flush=True)
for tool in ("unit_times.py", "unit_simds.py"):
    subprocess.run([sys.executable, str(ROOT / "tools" / tool), str(out)])
