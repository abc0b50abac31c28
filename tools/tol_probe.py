#!/usr/bin/env python3
"""The 5000-sweep solve at 16384x8192 in the tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE), timed; DMG_JACOBI_DEEP=0/1 and
DMG_DEEP_ROWS=<n> in the environment pick the form and the deep form's unit height.   python tools/tol_probe.py [survey|ocean]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch  # noqa: F401
from demiurge_amd import capi, synth
terrain = sys.argv[1] if len(sys.argv) > 1 else "survey"
W, H = 16384, 8192
ctx = capi.Context(0)
prm = capi.default_ocean_params()
prm.quirks |= capi.QUIRK_JACOBI_TOLERANCE
o = capi.Ocean(ctx, capi.make_grid(W, H), params=prm)
o.set_terrain(synth.terrain(W, H) if terrain == "survey" else -np.ones((H, W), np.float32))
o.set_divw(synth.divw(W, H))
best = 1e9
for rep in range(4):
    o.zero_pressure(); ctx.sync(); t0 = time.perf_counter(); o.jacobi_sweeps(5000); ctx.sync(); dt = time.perf_counter() - t0
    if rep: best = min(best, dt)
print("tolerance rendering, %s, DEEP=%s ROWS=%s: %.1f ms per solve = %.0f Mcell-steps/s (%s)" %
      (terrain, os.environ.get("DMG_JACOBI_DEEP", "-"), os.environ.get("DMG_DEEP_ROWS", "-"), best * 1e3, W * H * 5000 / best / 1e6, o.kernel_name()), flush=True)
