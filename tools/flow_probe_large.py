"""Flow-accumulation stage timings (DMG_FLOW_TIMING) at a given size.  usage: python tools/flow_probe_large.py [W H]"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["DMG_FLOW_TIMING"] = "1"
from demiurge_amd import capi, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
ctx = capi.Context(0)
g = capi.make_grid(W, H)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
ctx.flow_accumulate(g, h, 0.5, 0.5)
t0 = time.perf_counter()
ctx.flow_accumulate(g, h, 0.5, 0.5)
print("flow %dx%d: %.1f ms host-to-host" % (W, H, (time.perf_counter() - t0) * 1e3))
