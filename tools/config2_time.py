import sys, time, numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth
ctx = capi.Context()
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 2, 0.5, 1.0, 1.0); ta = time.perf_counter() - t0
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 12, 0.5, 1.0, 1.0); tb = time.perf_counter() - t0
print("per round fresh %.2f ms" % ((tb - ta) / 10 * 1e3))
t0 = time.perf_counter(); ctx.erosion_run(g, h, 50, 1000, 0.5, 1.0, 1.0); print("1000 iterations %.2f s" % (time.perf_counter() - t0))
