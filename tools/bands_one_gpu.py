"""The 16384x8192 solve split into n latitude bands that all live on GPU 0 (dmg_multi, peer copies): what banding itself
costs -- ghost-row redundancy, small launches, the exchange's synchronisation -- before any xGMI traffic.
usage: python tools/bands_one_gpu.py [sweeps [transport]]      transport: peer (default) | loopback (RCCL to the process's own rank) | store (direct stores)
Both exchange orders for every split: "overlap" puts the exchange on each band's second (high-priority) stream between the
two halves of the block's last launch.  All bands share one GPU here, so 2 n streams compete for its hardware queues:
GPU_MAX_HW_QUEUES is raised to 8 unless the caller has set it (with HIP's default of 4 the streams of different bands are
multiplexed onto one queue and serialise)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, ".")
from demiurge_amd import capi, synth

W, H = 16384, 8192
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
grid = capi.make_grid(W, H)
terrain, divw = synth.terrain(W, H), synth.divw(W, H)
transport = {"loopback": capi.MULTI_RCCL_LOOPBACK, "store": capi.MULTI_STORE}.get(sys.argv[2] if len(sys.argv) > 2 else "peer", capi.MULTI_PEER)
for n, ghost, order in ((1, 32, 0), (2, 32, 0), (2, 32, 1), (2, 64, 0), (2, 64, 1), (4, 32, 0), (4, 32, 1), (4, 64, 0), (4, 64, 1), (8, 32, 0), (8, 32, 1),
                        (8, 64, 0), (8, 64, 1), (8, 128, 0), (8, 128, 1)):
    flags = transport | (capi.MULTI_OVERLAP_LAST_LAUNCH if order else capi.MULTI_EXCHANGE_THEN_SWEEP)
    if order and n >= 8 and transport != capi.MULTI_RCCL_LOOPBACK:
        # more than four bands of one process on ONE device in the overlapped order: the library itself falls back to
        # exchange-then-sweep (9+ streams of one device that wait on each other are time-sliced by its scheduler:
        # profiles/r04_peer_stall.log) -- the line would repeat the one above
        print("bands %d ghost %3d order overlap: the library runs exchange-then-sweep here (DMG_MULTI_WARN_SHARED_GPU_FALLBACK)" % (n, ghost), flush=True)
        continue
    m = capi.Multi(grid, devices=[0] * n, ghost=ghost, params=capi.default_ocean_params(), flags=flags)
    m.set_terrain(terrain); m.set_divw(divw)
    m.jacobi(sweeps, zero_first=True); m.sync()
    t0 = time.perf_counter()
    m.jacobi(sweeps, zero_first=True); m.sync()
    dt = time.perf_counter() - t0
    print("bands %d ghost %3d order %s: %.1f ms  %.0f Mcell-steps/s" % (n, ghost, "overlap" if order else "exchange-then-sweep", dt * 1e3, W * H * sweeps / dt / 1e6), flush=True)
    m.close()
