"""Times the gradient-noise generator (device-resident: dmg_ocean_generate_terrain) per mode at 16384x8192."""
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = W // 2
ctx = capi.Context(0)
oc = capi.Ocean(ctx, capi.make_grid(W, H))
for mode in capi.NOISE_MODES:
    p = capi.noise_params(mode, seed=3, warp=0.7)
    oc.generate_terrain(p); ctx.sync()
    ctx.timer_start()
    for _ in range(3):
        oc.generate_terrain(p)
    ms = ctx.timer_stop() / 3
    print("%-9s %dx%d: %.2f ms incl. neighbour codes + row classes (%.1f Gtexel/s)" % (mode, W, H, ms, W * H / ms / 1e6))
