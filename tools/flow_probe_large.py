import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
os.environ["DMG_FLOW_TIMING"] = "1"
from demiurge_amd import capi, synth
W, H = 16384, 8192
ctx = capi.Context(0)
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.flow_accumulate(g, h, 0.5, 0.5)
t0 = time.perf_counter(); f = ctx.flow_accumulate(g, h, 0.5, 0.5); print("flow 16384x8192: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
