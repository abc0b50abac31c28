"""Where an erosion round at 16384x8192 spends its time: DMG_FLOW_TIMING=1 prints the flow solve's phases (GPU box).
usage: python tools/flow_timing.py [W]   (H = W / 2; the config-5 terrain of tools/config_lines.py)"""
import os, sys, time
import numpy as np
os.environ.setdefault("DMG_FLOW_TIMING", "1")
sys.path.insert(0, ".")
from demiurge_amd import capi, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = W // 2
ctx = capi.Context()
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
e = capi.Erosion(ctx, g, h, N=50, exponent=0.5, slope_exponent=1.0, factor=1.0, lakes=0)
for i in range(3):
    t0 = time.perf_counter()
    e.rounds(1)
    ctx.sync()
    print("round %d: %.3f s" % (i, time.perf_counter() - t0), flush=True)
e.close()
