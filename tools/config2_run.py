"""Config 2's loop for a profiler: python tools/config2_run.py [iterations] (4096x2048 stream-power erosion)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from demiurge_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = capi.Context()
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, n, 0.5, 1.0, 1.0); print("%d iterations %.3f s" % (n, time.perf_counter() - t0))
