#!/usr/bin/env python3
"""Times BASELINE config 2 pieces on one GPU: one flow accumulation and a few rounds of the stream-power erosion
loop (cpufilter::run) at 4096x2048, host buffers in and out."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from demiurge_amd import capi, synth

W, H = 4096, 2048
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = capi.Context(0)
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.flow_accumulate(g, h, 0.5, 0.5)
t0 = time.perf_counter(); f = ctx.flow_accumulate(g, h, 0.5, 0.5); t1 = time.perf_counter()
print("flow accumulation %dx%d: %.1f ms (drained cells %d, max area^0.5 %.3f)" % (W, H, (t1 - t0) * 1e3, (f > 0).sum(), f.max()))
t0 = time.perf_counter(); out = ctx.erosion_run(g, h, 50, iters, 0.5, 1.0, 1.0); t1 = time.perf_counter()
print("erosion loop: %d rounds in %.2f s = %.1f ms per round -> %.1f Mcell-iters/s" % (iters, t1 - t0, (t1 - t0) / iters * 1e3, W * H * iters / (t1 - t0) / 1e6))
