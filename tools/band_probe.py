#!/usr/bin/env python3
"""Times the Jacobi sweeps of ONE interior latitude band of the 16384x8192 grid on one GPU (no halo exchange: the
numbers are garbage, the launch geometry is that of rank r of n).  Used to tune tile shapes for the N-GPU run."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from demiurge_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ghost = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sweeps = 1600
W, H = 16384, 8192
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
oc = capi.Ocean(ctx, capi.make_grid(W, H), band_rank=n // 2, band_count=n, ghost=ghost)
oc.set_terrain(synth.terrain(W, H))
oc.set_divw(synth.divw(W, H))
for split in (False, True):          # whole blocks, then blocks whose last launch is issued as edge tiles + interior tiles
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        done = 0
        while done < sweeps:
            if split:
                oc.jacobi_sweeps_begin(ghost); oc.jacobi_sweeps_end()
            else:
                oc.jacobi_sweeps(ghost)
            done += ghost
        ctx.sync(); dt = time.perf_counter() - t0
    print("bands %d ghost %d %s: %.1f us per 8 sweeps of the band, %.1f us per block -> %.0f Mcell-steps/s per GPU (x%d = %.0f)" %
          (n, ghost, "split last launch" if split else "whole launches  ", dt / sweeps * 8 * 1e6, dt / sweeps * ghost * 1e6,
           W * (H / n) * sweeps / dt / 1e6, n, W * H * sweeps / dt / 1e6))
