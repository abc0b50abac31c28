#!/usr/bin/env python3
"""Times the Jacobi sweeps of EVERY latitude band of an n-way split of the 16384x8192 grid on one GPU (no halo exchange: the
numbers are garbage, the launch geometry is that of rank r of n) -- the kernel's share of what an n-GPU run can reach before a
byte crosses xGMI (every band, polar ones too, not the middle one).
    python tools/band_probe.py [n ...] [--ghost 32] [--tolerance] [--split]
"""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from demiurge_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int, nargs="*", default=[8, 4, 2])
ap.add_argument("--ghost", type=int, default=32)
ap.add_argument("--sweeps", type=int, default=1600)
ap.add_argument("--tolerance", action="store_true", help="DMG_QUIRK_JACOBI_TOLERANCE")
ap.add_argument("--split", action="store_true", help="also the split-last-launch order of the middle band")
ap.add_argument("--bands", default="", help="comma list of ranks (default: all)")
a = ap.parse_args()
W, H = 16384, 8192
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
img, divw = synth.terrain(W, H), synth.divw(W, H)
ghost, sweeps = a.ghost, a.sweeps
full_us = None


def run(oc, split):
    best = 1e30
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        done = 0
        while done < sweeps:
            if split:
                oc.jacobi_sweeps_begin(ghost); oc.jacobi_sweeps_end()
            else:
                oc.jacobi_sweeps(ghost)
            done += ghost
        ctx.sync(); best = min(best, time.perf_counter() - t0)
    return best


for n in a.n:
    ranks = [int(x) for x in a.bands.split(",")] if a.bands else list(range(n))
    worst = 0.0
    for r in ranks:
        prm = capi.default_ocean_params()
        if a.tolerance:
            prm.quirks |= capi.QUIRK_JACOBI_TOLERANCE
        oc = capi.Ocean(ctx, capi.make_grid(W, H), params=prm, band_rank=r, band_count=n, ghost=ghost)
        oc.set_terrain(img)
        oc.set_divw(divw)
        modes = [False] + ([True] if a.split and r == n // 2 else [])
        for split in modes:
            dt = run(oc, split)
            us8 = dt / sweeps * 8 * 1e6
            if not split:
                worst = max(worst, us8)
            print("bands %d band %d ghost %d %s%s: %.1f us per 8 sweeps of the band, %.1f us per block -> %.0f Mcell-steps/s per GPU (%s)" %
                  (n, r, ghost, "split last launch" if split else "whole launches  ", " TOLERANCE" if a.tolerance else "", us8, dt / sweeps * ghost * 1e6,
                   W * (H / n) * sweeps / dt / 1e6, oc.kernel_name()), flush=True)
        oc.close()
    print("bands %d: slowest band %.1f us per 8 sweeps -> x%d = %.0f Mcell-steps/s before any exchange" % (n, worst, n, W * H * 8 / worst), flush=True)
