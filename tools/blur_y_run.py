#!/usr/bin/env python3
"""A few (0, r) blur passes at 16384x8192 for a profiler: python3 tools/blur_y_run.py [r] [passes]"""
import os, sys
import numpy as np
os.environ.setdefault("DMG_TEST_HOOKS", "1")
sys.path.insert(0, ".")
import torch  # noqa: F401
from demiurge_amd import capi, synth
r = float(sys.argv[1]) if len(sys.argv) > 1 else 0.12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H = 16384, 8192
ctx = capi.Context(0)
h = (synth.terrain(W, H) * 100 + synth.lcg_uniform(W * H, 5).reshape(H, W) * 3).astype(np.float32)
f = capi.Field(ctx, capi.make_grid(W, H), 0, 1, 0, data=h)
for _ in range(n):
    f.blur_pass(r, True)
ctx.sync()
f.close(); ctx.close()
