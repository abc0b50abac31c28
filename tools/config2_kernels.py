"""config 2's erosion rounds under rocprofv3 --kernel-trace --stats: which kernels a round of the lake graph spends its device time
in (200 rounds at 4096x2048 after 100 warm-up rounds: the lakes have thinned out as in the config's 1000).
usage: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2k -o x -- python3 tools/config2_kernels.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth
ctx = capi.Context()
W, H = 4096, 2048
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.erosion_run(g, h, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 300, 0.5, 1.0, 1.0)
