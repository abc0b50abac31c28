"""Worker for the multi-process latitude-band tests (spawned with torch.multiprocessing, backend gloo).

mode "oracle": CPU only.  The band's compute is the oracle (tests may use it), wrapped in FakeOcean, which has the
               same halo()/jacobi_sweeps() contract as demiurge_amd.capi.Ocean; this checks the exchange protocol
               of demiurge_amd/bands.py (which rows go where, ordering, no deadlock) without a GPU.
mode "hip":    every rank drives the real HIP library on GPU 0 (several ranks share the one GPU of a test box),
               slabs staged through the host because gloo cannot send device memory.
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


class FakeOcean:
    """Band of the Jacobi solve computed by the oracle on host arrays, ghost rows included."""

    def __init__(self, oracle, W, H, rank, count, ghost, img, divw):
        self.oracle, self.W, self.H, self.ghost = oracle, W, H, ghost
        self.y0, self.y1 = H * rank // count, H * (rank + 1) // count
        self.gs = ghost if rank > 0 else 0
        self.gn = ghost if rank < count - 1 else 0
        self.g = oracle.grid(W, H)
        self.img, self.divw = img, divw
        lo, hi = self.y0 - self.gs, self.y1 + self.gn
        self.lo, self.hi = lo, hi
        self.p = np.zeros((hi - lo, W), np.float32)

    def zero_pressure(self):
        self.p[:] = 0

    def halo(self, field, side, rows):
        own = self.y1 - self.y0
        if side == 0 and self.gs:
            return self.p[self.gs:self.gs + rows], self.p[self.gs - rows:self.gs], rows * self.W * 4
        if side == 1 and self.gn:
            return self.p[self.gs + own - rows:self.gs + own], self.p[self.gs + own:self.gs + own + rows], rows * self.W * 4
        return None, None, 0

    def jacobi_sweeps(self, n):
        assert n <= self.ghost or (self.gs == 0 and self.gn == 0)
        # embed the band (with ghosts) in a full grid: rows further than n away cannot influence owned rows in n sweeps
        full = np.full((self.H, self.W), np.float32(1e30))
        full[self.lo:self.hi] = self.p
        out = self.oracle.ocean_jacobi(self.g, self.img, self.divw, n, p0=full, rows=True, nthreads=2)
        self.p[self.gs:self.gs + (self.y1 - self.y0)] = out[self.y0:self.y1]

    def owned(self):
        return self.p[self.gs:self.gs + (self.y1 - self.y0)]


def worker(rank, world, port, mode, W, H, sweeps, ghost, outdir, step=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from demiurge_amd import bands, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img, divw = synth.terrain(W, H), synth.divw(W, H)
    if mode == "oracle":
        import oracle
        oc = FakeOcean(oracle, W, H, rank, world, ghost, img, divw)
        exch = bands.HaloExchanger(dist, rank, world)
        bands.jacobi_solve(oc, exch, sweeps, ghost)
        mine = oc.owned()
    else:
        import torch
        from demiurge_amd import capi
        torch.cuda.set_device(0)
        # the library must run on the stream torch orders its copies / collectives on (as bench.py does)
        ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        oc = capi.Ocean(ctx, capi.make_grid(W, H), band_rank=rank, band_count=world, ghost=ghost)
        oc.set_terrain(img)
        oc.set_divw(divw)
        exch = bands.HaloExchanger(dist, rank, world, torch.device("cuda", 0))
        lo, hi = H * rank // world, H * (rank + 1) // world
        if step is None:
            bands.jacobi_solve(oc, exch, sweeps, ghost)
            mine = oc.get(want=("pressure",))["pressure"][lo:hi]
        else:                                           # full outer ocean step(s): (n_outer, jacobi_iters, diffusion_iters)
            oc.close()
            prm = capi.default_ocean_params()
            prm.jacobi_iters, prm.diffusion_iters = step[1], step[2]
            oc = capi.Ocean(ctx, capi.make_grid(W, H), params=prm, band_rank=rank, band_count=world, ghost=ghost)
            oc.set_terrain(img)
            v0 = synth.velocity(W, H)
            v0[img > 0] = 0
            oc.set_velocity(v0)
            bands.ocean_step(oc, exch, step[0])
            r = oc.get(want=("v", "pressure"))
            mine = np.concatenate([r["v"][lo:hi], r["pressure"][lo:hi, :, None]], axis=2)
        oc.close()
        ctx.close()
    np.save(os.path.join(outdir, "band%d.npy" % rank), mine)
    dist.barrier()
    dist.destroy_process_group()
