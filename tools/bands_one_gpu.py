"""The 16384x8192 solve split into n latitude bands that all live on GPU 0 (dmg_multi, peer copies): what banding itself
costs -- ghost-row redundancy, small launches, the exchange's synchronisation -- before any xGMI traffic.
usage: python tools/bands_one_gpu.py [sweeps [transport]]      transport: peer (default) | loopback (RCCL to the process's own rank)
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
transport = capi.MULTI_RCCL_LOOPBACK if (len(sys.argv) > 2 and sys.argv[2] == "loopback") else capi.MULTI_PEER
for n, ghost, order in ((1, 32, 0), (2, 32, 0), (2, 32, 1), (2, 64, 0), (2, 64, 1), (4, 32, 0), (4, 32, 1), (4, 64, 0), (4, 64, 1), (8, 32, 0), (8, 32, 1),
                        (8, 64, 0), (8, 64, 1), (8, 128, 0), (8, 128, 1)):
    flags = transport | (capi.MULTI_OVERLAP_LAST_LAUNCH if order else capi.MULTI_EXCHANGE_THEN_SWEEP)
    if order and n >= 8 and transport == capi.MULTI_PEER:
        # 16 streams on ONE device with event waits between all of them: the runtime's hardware queues are oversubscribed and
        # every wait between two bands' streams costs a scheduling quantum (measured 0.7 .. 16 s for this line, with 8, 16 or 24
        # GPU_MAX_HW_QUEUES alike).  An artefact of rehearsing eight devices on one: each real device holds two streams.  The
        # loopback transport is the rehearsal that means something at 8 bands.
        print("bands %d ghost %3d order overlap: skipped (peer copies between 16 streams of one device oversubscribe its hardware queues)" % (n, ghost), flush=True)
        continue
    m = capi.Multi(grid, devices=[0] * n, ghost=ghost, params=capi.default_ocean_params(), flags=flags)
    m.set_terrain(terrain); m.set_divw(divw)
    m.jacobi(sweeps, zero_first=True); m.sync()
    t0 = time.perf_counter()
    m.jacobi(sweeps, zero_first=True); m.sync()
    dt = time.perf_counter() - t0
    print("bands %d ghost %3d order %s: %.1f ms  %.0f Mcell-steps/s" % (n, ghost, "overlap" if order else "exchange-then-sweep", dt * 1e3, W * H * sweeps / dt / 1e6), flush=True)
    m.close()
