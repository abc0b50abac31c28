#!/usr/bin/env python3
"""Times the passes of one outer ocean step (OceanCurrents::run body) on one GPU at BASELINE config 3/4 sizes."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from demiurge_amd import capi, synth

import itertools
for (W, H), fast in itertools.product(((8192, 4096), (16384, 8192)), (False, True)):
    ctx = capi.Context(0)
    prm = capi.default_ocean_params()
    if fast:
        prm.quirks |= capi.QUIRK_FAST_TRANSCENDENTALS     # advect and divergence on the hardware's transcendentals (within 1e-5)
    o = capi.Ocean(ctx, capi.make_grid(W, H), params=prm)
    img = synth.terrain(W, H)
    o.set_terrain(img)
    v = synth.velocity(W, H); v[img > 0] = 0
    o.set_velocity(v); del v
    def t(f, n=1):
        f(); ctx.sync(); t0 = time.perf_counter()
        for _ in range(n): f()
        ctx.sync(); return (time.perf_counter() - t0) / n * 1e3
    ta = t(o.advect, 3); td = t(lambda: o.diffusion(50)); tv = t(o.divergence, 3)
    tj = t(lambda: o.jacobi(5000)); ts = t(o.subdiv, 3)
    tot = ta + td + tv + tj + ts
    cells = W * H
    print(("tolerance mode " if fast else "exact (default) ") + "%dx%d  advect %.2f ms (%.0f GB/s alg)  diffusion x50 %.2f ms (%.0f GB/s alg)  divergence %.2f ms (%.0f GB/s alg)  "
          "jacobi x5000 %.1f ms  subDiv %.2f ms (%.0f GB/s alg)  | outer step %.1f ms" %
          (W, H, ta, 28 * cells / ta / 1e6, td, 50 * 20 * cells / td / 1e6, tv, 16 * cells / tv / 1e6, tj, ts, 24 * cells / ts / 1e6, tot))
    o.close(); ctx.close()
