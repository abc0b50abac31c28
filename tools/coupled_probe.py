#!/usr/bin/env python3
"""BASELINE config 5 pieces that exist (uplift + stream-power erosion round, climate step) at 16384x8192 on one GPU."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from demiurge_amd import capi, synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 8192)
ctx = capi.Context(0)
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
t0 = time.perf_counter(); out = ctx.erosion_run(g, h, 50, 2, 0.5, 1.0, 1.0); t1 = time.perf_counter()
print("erosion: 2 rounds at %dx%d in %.2f s incl. PCIe = %.0f ms per round" % (W, H, t1 - t0, (t1 - t0) / 2 * 1e3))
T0 = np.full((H, W), 50, np.float32)
ctx.temperature_steps(g, T0, h, 0, 1)
t0 = time.perf_counter(); T = ctx.temperature_steps(g, T0, h, 0, 1); ta = time.perf_counter() - t0
t0 = time.perf_counter(); T = ctx.temperature_steps(g, T0, h, 0, 41); tb = time.perf_counter() - t0
print("climate: %.2f ms per step at %dx%d (%.0f GB/s algorithmic at 12 B/cell); T range %.2f..%.2f" %
      ((tb - ta) / 40 * 1e3, W, H, 12 * W * H / ((tb - ta) / 40) / 1e9, T.min(), T.max()))
