"""BASELINE config 5's coupled iteration (dmg_world_step: a tectonics step, an erosion round, ten climate steps) at 16384x8192 under
rocprofv3 --kernel-trace --stats: which kernels the iteration's device time goes to.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wk -o x -- python3 tools/world_kernels.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth
ctx = capi.Context()
W, H = 16384, 8192
g = capi.make_grid(W, H)
terr = (synth.terrain(W, H) * 100).astype(np.float32)
T0 = np.full((H, W), 50, np.float32)
w = capi.World(ctx, g, terr, T0)
for _ in range(4):
    w.step()
ctx.sync()
w.close()
