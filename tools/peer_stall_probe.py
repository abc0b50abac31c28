"""One banded solve with all bands on GPU 0 (dmg_multi, one process): how long it takes, for a transport / order / band count.
usage: python tools/peer_stall_probe.py <bands> <transport: peer|loopback|store> <order: plain|overlap> [sweeps [ghost [height]]]
Run it under different GPU_MAX_HW_QUEUES / DMG_MULTI_COMM_PRIORITY settings (read when HIP / the library initialise)."""
import os, sys, time
os.environ.setdefault("DMG_MULTI_FORCE_OVERLAP", "1")     # this probe measures the very case the library avoids
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiurge_amd import capi, synth

n = int(sys.argv[1]); transport = sys.argv[2]; order = sys.argv[3]
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 320
ghost = int(sys.argv[5]) if len(sys.argv) > 5 else 32
H = int(sys.argv[6]) if len(sys.argv) > 6 else 8192
W = 16384
grid = capi.make_grid(W, H)
tf = {"peer": capi.MULTI_PEER, "loopback": capi.MULTI_RCCL_LOOPBACK, "store": getattr(capi, "MULTI_STORE", 4)}[transport]
flags = tf | (capi.MULTI_OVERLAP_LAST_LAUNCH if order == "overlap" else capi.MULTI_EXCHANGE_THEN_SWEEP)
m = capi.Multi(grid, devices=[0] * n, ghost=ghost, params=capi.default_ocean_params(), flags=flags)
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
m.jacobi(sweeps, zero_first=True); m.sync()
t0 = time.perf_counter()
m.jacobi(sweeps, zero_first=True); m.sync()
dt = time.perf_counter() - t0
st = m.stats()
print("hwq=%s prio=%s bands %d %s %s ghost %d: %d sweeps in %.1f ms (%.0f Mcell-steps/s), %d exchanges, %.2f ms on band 0's exchange stream" %
      (os.environ.get("GPU_MAX_HW_QUEUES", "-"), os.environ.get("DMG_MULTI_COMM_PRIORITY", "-"), n, transport, order, ghost, sweeps, dt * 1e3,
       W * H * sweeps / dt / 1e6, st["exchanges"], st["exchange_ms"]), flush=True)
m.close()
