"""The stream-power erosion round at 16384x8192 on a terrain with noise (tens of millions of lakes: config 2's workload at config 5's
size) under rocprofv3 --kernel-trace --stats: kernels and the gaps between them (host phases).
usage: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ek -o x -- python3 tools/erosion_big_kernels.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth
ctx = capi.Context()
W, H = 16384, 8192
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, 1, 0.5, 1.0, 1.0)
ctx.erosion_run(g, h, 50, 4, 0.5, 1.0, 1.0)
