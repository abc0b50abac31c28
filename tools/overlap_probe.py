"""One rank's timeline of a banded solve, on one GPU: a 16384 x (n * rows) grid as n latitude bands of `rows` rows (1024 = a band
of the 16384x8192 grid over 8 GPUs) driven by dmg_multi over the RCCL loopback transport -- all bands on ONE pair of streams,
so that what the exchange costs, and what the overlapped order hides of it, shows as it would on a rank that has the GPU to
itself.  Run it under `rocprofv3 --kernel-trace` and feed the trace to tools/overlap_trace.py.
usage: python tools/overlap_probe.py <order: plain|overlap> [rows per band [bands [sweeps [ghost]]]]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # the overlapped order needs eight hardware queues, asked for before HIP starts
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth

order = sys.argv[1] if len(sys.argv) > 1 else "overlap"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bands = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 640
ghost = int(sys.argv[5]) if len(sys.argv) > 5 else 32
W, H = 16384, rows * bands
grid = capi.make_grid(W, H)
flags = capi.MULTI_RCCL_LOOPBACK | (capi.MULTI_OVERLAP_LAST_LAUNCH if order == "overlap" else capi.MULTI_EXCHANGE_THEN_SWEEP)
m = capi.Multi(grid, devices=[0] * bands, ghost=ghost, flags=flags)
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
m.jacobi(sweeps, zero_first=True); m.sync()
t0 = time.perf_counter()
m.jacobi(sweeps, zero_first=True); m.sync()
dt = time.perf_counter() - t0
st = m.stats()
print("order %s: %d bands x %d rows, ghost %d, %d sweeps: %.2f ms, %.0f Mcell-steps/s; %d exchanges, %.3f ms on their stream" %
      (order, bands, rows, ghost, sweeps, dt * 1e3, W * H * sweeps / dt / 1e6, st["exchanges"], st["exchange_ms"]), flush=True)
m.close()
