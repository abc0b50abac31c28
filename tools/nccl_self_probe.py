"""One process, one GPU, backend nccl (= RCCL): two latitude bands of one grid exchange their halos with
ncclSend/ncclRecv to the process's own rank.  No second GPU is needed, yet the transfers run on RCCL's own stream, so
this exercises what the gloo tests cannot: the stream ordering of the overlapped solve (edge tiles -> sends/receives on
the RCCL stream -> interior tiles on the compute stream -> wait) and zero-copy sends from the library's device slabs.
Prints the result of the bitwise comparison with a single band and the time of a block with and without exchange.

    python tools/nccl_self_probe.py [W H sweeps ghost]
"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from demiurge_amd import bands, capi, synth  # noqa: E402

W, H, sweeps, ghost = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 1024, 200, 32)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
grid = capi.make_grid(W, H)
img, b = synth.terrain(W, H), synth.divw(W, H)
prm = capi.default_ocean_params()
prm.jacobi_kernel = capi.JACOBI_BLOCKED


class SelfExchanger(bands.HaloExchanger):
    """both bands live in this process: band 0's north slab goes to band 1's south ghosts and back, through RCCL"""

    def start(self, oceans, field, rows, buffer=None):      # the interface bands.jacobi_solve_pipelined drives
        if os.environ.get("DMG_PROBE_NO_EXCHANGE"):
            return [], [], oceans
        return self.start_pair(oceans[0], oceans[1], field, rows, buffer), [], oceans

    def exchange(self, oceans, field, rows):
        self.finish(self.start(oceans, field, rows))

    def start_pair(self, o0, o1, field, rows, buffer=None):
        halo = (lambda o, side: o.halo(field, side, rows)) if buffer is None else (lambda o, side: o.pressure_halo(buffer, side, rows))
        s0, r0, nb = halo(o0, capi.SIDE_NORTH)
        s1, r1, nb1 = halo(o1, capi.SIDE_SOUTH)
        assert nb == nb1 and nb > 0
        t = self._tensor
        ops = [dist.P2POp(dist.isend, t(s0, nb), 0), dist.P2POp(dist.irecv, t(r1, nb), 0),
               dist.P2POp(dist.isend, t(s1, nb), 0), dist.P2POp(dist.irecv, t(r0, nb), 0)]
        return dist.batch_isend_irecv(ops)


ex = SelfExchanger(dist, 0, 1, device=torch.device("cuda", 0))


def solve(overlap, exchange=True):
    oc = [capi.Ocean(ctx, grid, params=prm, band_rank=r, band_count=2, ghost=ghost) for r in range(2)]
    for o in oc:
        o.set_terrain(img); o.set_divw(b); o.zero_pressure()
    if overlap == "pipelined":
        ctx.sync(); t0 = time.perf_counter()
        bands.jacobi_solve_pipelined(oc, ex, sweeps, ghost, zero_first=False)
        t_host = time.perf_counter() - t0
        ctx.sync(); dt = time.perf_counter() - t0
        if os.environ.get("DMG_PROBE_HOST"):
            print("  pipelined: host issued everything after %.0f us per block, GPU done after %.0f us per block" %
                  (t_host / ((sweeps + ghost - 1) // ghost) * 1e6, dt / ((sweeps + ghost - 1) // ghost) * 1e6))
        out = {"pressure": np.zeros((H, W), np.float32)}
        for o in oc:
            o.get(want=("pressure",), out=out); o.close()
        return out["pressure"], dt
    def xch():
        if exchange:
            for w in ex.start_pair(oc[0], oc[1], capi.F_PRESSURE, ghost):
                w.wait()
    xch()
    ctx.sync(); t0 = time.perf_counter()
    done = 0
    while done < sweeps:
        n = min(ghost, sweeps - done)
        last = done + n >= sweeps
        if overlap and not last:
            for o in oc:
                o.jacobi_sweeps_begin(n)
            works = ex.start_pair(oc[0], oc[1], capi.F_PRESSURE, ghost) if exchange else []
            for o in oc:
                o.jacobi_sweeps_end()
            for w in works:
                w.wait()
        else:
            for o in oc:
                o.jacobi_sweeps(n)
            if not last:
                xch()
        done += n
    ctx.sync(); dt = time.perf_counter() - t0
    out = {"pressure": np.zeros((H, W), np.float32)}
    for o in oc:
        o.get(want=("pressure",), out=out); o.close()
    return out["pressure"], dt


one = capi.Ocean(ctx, grid, params=prm)
one.set_terrain(img); one.set_divw(b); one.zero_pressure(); one.jacobi_sweeps(sweeps)
ref = one.get(want=("pressure",))["pressure"]; one.close()
solve(True)                                       # warm-up: RCCL communicator, tile lists, tuner
blocks = (sweeps + ghost - 1) // ghost
solve("pipelined")
for name, ov, xc in (("exchange then sweep", False, True), ("overlapped last launch", True, True), ("pipelined edge chain", "pipelined", True),
                     ("no exchange (garbage)", False, False)):
    best = 1e9
    for _ in range(3):
        p, dt = solve(ov, xc)
        best = min(best, dt)
    ok = "" if not xc else ("  bit-identical to one band: %s" % np.array_equal(p, ref))
    print("%-24s %.1f us per block of %d sweeps (two bands on one GPU)%s" % (name, best / blocks * 1e6, ghost, ok))
dist.destroy_process_group()
