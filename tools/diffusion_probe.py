"""Times the velocity-diffusion pass (50 sweeps, OceanCurrents.cpp:701-707) on one GPU.

    python tools/diffusion_probe.py [W]            blocked kernel (default build)
    DMG_DIFFUSION_SINGLE=1 python tools/...        one sweep per launch
    DMG_DTB_ROWS=4|6|8 python tools/...            force the tile height
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = W // 2
ctx = capi.Context(0)
oc = capi.Ocean(ctx, capi.make_grid(W, H))
oc.set_terrain(synth.terrain(W, H))
oc.set_velocity(synth.velocity(W, H))
oc.diffusion(50); ctx.sync()
reps = 5
ctx.timer_start()
for _ in range(reps):
    oc.diffusion(50)
ms = ctx.timer_stop() / reps
print("%dx%d  50 diffusion sweeps: %.3f ms  (%.1f us per sweep, %.0f Mcell-sweeps/s)  single=%s rows=%s" %
      (W, H, ms, ms * 1e3 / 50, W * H * 50 / ms / 1e3, os.environ.get("DMG_DIFFUSION_SINGLE", "0"), os.environ.get("DMG_DTB_ROWS", "auto")))
