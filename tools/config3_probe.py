#!/usr/bin/env python3
"""Config 3's pressure solve (8192x4096) beside a grid of the same cell count (16384x2048): 5000 sweeps each, and one warm launch's
units at 8192x4096 (DMG_ST_PROF) through tools/unit_times.py and tools/unit_simds.py.
    DMG_TEST_HOOKS=1 python tools/config3_probe.py [W H]"""
import os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
W0, H0 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 4096)
out = ROOT / "gpurun_out" / ("config3_units_%dx%d.csv" % (W0, H0))
os.environ["DMG_ST_PROF"] = str(out)
os.environ["DMG_ST_PROF_SKIP"] = "400"
os.environ.setdefault("DMG_TEST_HOOKS", "1")
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
from demiurge_amd import capi, synth
ctx = capi.Context(0)
for W, H in ((W0, H0), (16384, 2048), (W0, H0)):
    oc = capi.Ocean(ctx, capi.make_grid(W, H))
    oc.set_terrain(synth.terrain(W, H)); oc.set_divw(synth.divw(W, H))
    oc.jacobi(1000); ctx.sync()
    t0 = time.perf_counter(); oc.jacobi(5000); ctx.sync(); dt = time.perf_counter() - t0
    print("%dx%d: 5000 sweeps %.2f ms = %.1f us per 8-sweep launch, %.0f Mcell-steps/s" % (W, H, dt * 1e3, dt / 625 * 1e6, W * H * 5000 / dt / 1e6), flush=True)
    oc.close()
ctx.close()
if out.exists():
    for tool in ("unit_times.py", "unit_simds.py"):
        subprocess.run([sys.executable, str(ROOT / "tools" / tool), str(out)])
