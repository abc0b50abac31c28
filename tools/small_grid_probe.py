"""Jacobi solve time at small and medium grids (launch/tile-latency bound regime).  DMG_TB_SHAPE=<R>x<NW> picks the tile."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
for W, H in ((1024, 512), (2048, 1024), (4096, 2048), (8192, 4096)):
    oc = capi.Ocean(ctx, capi.make_grid(W, H))
    oc.set_terrain(synth.terrain(W, H)); oc.set_divw(synth.divw(W, H))
    oc.jacobi(1000); ctx.sync()
    t0 = time.perf_counter(); oc.jacobi(5000); ctx.sync(); dt = time.perf_counter() - t0
    print("%s %dx%d: 5000 sweeps %.2f ms = %.1f us per 8-sweep launch, %.0f Mcell-steps/s"
          % (os.environ.get("DMG_TB_SHAPE", "default"), W, H, dt * 1e3, dt / 625 * 1e6, W * H * 5000 / dt / 1e6))
    oc.close()
