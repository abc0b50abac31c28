"""Latitude-band driver: one process per GPU, pressure ghost rows refreshed with point-to-point exchanges.

The reference has no distributed piece; this is the only communication the backend
introduces.  A band owns H/n contiguous rows x full W, so the longitude wrap and the pole reflection stay
band-local.  Every `ghost` sweeps each rank sends its `ghost` outermost owned rows to the
neighbouring rank and receives that rank's rows into its ghost slab: the same bytes as a 1-row exchange per
sweep, in 1/ghost as many messages (xGMI is latency-bound at 64 KiB).  Results are bit-identical to one band.

Transport: `torch.distributed` P2P -- backend "nccl" (= RCCL over xGMI) sends straight from / into the library's
device slabs (zero copy); with backend "gloo" (CPU tests, or several ranks sharing one GPU) slabs are staged
through host tensors.  In the Jacobi solve the exchange for the next block of sweeps overlaps the interior tiles of the
current block's last launch (jacobi_solve).

Stream contract: create the dmg context on the stream torch issues its copies / collectives on
(`capi.Context(dev, stream=torch.cuda.current_stream().cuda_stream)`), so kernels, halo sends and halo receives
are ordered by the stream itself and no host synchronisation is needed between sweep blocks.
"""
import numpy as np


class _DevSlab:
    """__cuda_array_interface__ view of `nbytes` of library-owned device memory."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = dict(shape=(nbytes // 4,), typestr="<f4", data=(int(ptr), False), version=2)


class HaloExchanger:
    """modes: zero-copy aliases of the library's slabs (nccl), torch-owned device bounce buffers filled by
    device-to-device copies (`bounce`: DMG_HALO_BOUNCE=1, or automatically when torch cannot alias the slabs),
    host staging on top of either (any backend but nccl)."""

    def __init__(self, dist, rank, world, device=None):
        import os
        import torch
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.staged = dist.get_backend() != "nccl"
        self.bounce = os.environ.get("DMG_HALO_BOUNCE") == "1"
        self._alias = {}
        self._bounce = {}

    def _tensor(self, ptr, nbytes):
        """torch view of a slab: device alias for HIP memory, numpy-backed view when `ptr` is a host ndarray."""
        if isinstance(ptr, np.ndarray):
            return self.torch.from_numpy(ptr)
        key = (ptr, nbytes)
        if key not in self._alias:
            self._alias[key] = self.torch.as_tensor(_DevSlab(ptr, nbytes), device=self.device)
        return self._alias[key]

    def _bounce_pair(self, key, nbytes):
        if key not in self._bounce:
            mk = lambda: self.torch.empty(nbytes // 4, dtype=self.torch.float32, device=self.device)
            self._bounce[key] = (mk(), mk())
        return self._bounce[key]

    def start(self, ocean, field, rows):
        """Post the sends and receives that refresh `rows` ghost rows of `field` on both sides of this rank's band and
        return a token for finish().  With backend nccl the transfers run on RCCL's stream behind everything issued so
        far on the compute stream; whatever the caller launches before finish() overlaps them."""
        dist, torch = self.dist, self.torch
        ops, post = [], []
        for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):     # DMG_SIDE_SOUTH, DMG_SIDE_NORTH
            if peer < 0 or peer >= self.world:
                continue
            send, recv, nbytes = ocean.halo(field, side, rows)
            if isinstance(send, np.ndarray):
                ts, tr = self._tensor(send, nbytes), self._tensor(recv, nbytes)
            else:
                if not self.bounce:
                    try:
                        ts, tr = self._tensor(send, nbytes), self._tensor(recv, nbytes)
                    except Exception as e:   # torch build that cannot wrap foreign device memory: copy through its own
                        import sys
                        print("HaloExchanger: aliasing device slabs failed (%s); using bounce buffers" % e, file=sys.stderr)
                        self.bounce = True
                if self.bounce:
                    ts, tr = self._bounce_pair((field, side, nbytes), nbytes)
                    ocean.ctx.copy_device(ts.data_ptr(), send, nbytes)      # same stream as the sweeps and as torch
                    post.append(("d2d", recv, tr, nbytes))
            if self.staged and ts.is_cuda:
                hs, hr = ts.cpu(), torch.empty(ts.shape, dtype=ts.dtype)
                post.insert(0, ("h2d", tr, hr, nbytes))
                ts, tr = hs, hr
            ops.append(dist.P2POp(dist.isend, ts, peer))
            ops.append(dist.P2POp(dist.irecv, tr, peer))
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, post, ocean

    def finish(self, token):
        """Wait for the transfers of start() (nccl: the compute stream waits, not the host) and land staged copies."""
        works, post, ocean = token
        for w in works:
            w.wait()
        for kind, dst, src, nbytes in post:
            if kind == "h2d":
                dst.copy_(src)
            else:
                ocean.ctx.copy_device(dst, src.data_ptr(), nbytes)

    def exchange(self, ocean, field, rows):
        """Refresh `rows` ghost rows of `field` on both sides of this rank's band."""
        self.finish(self.start(ocean, field, rows))


def _overlap_enabled(ocean):
    import os
    return os.environ.get("DMG_HALO_OVERLAP", "1") != "0" and hasattr(ocean, "jacobi_sweeps_begin")


def jacobi_solve(ocean, exch, sweeps, ghost, zero_first=True, field=0):
    """OceanCurrents::jacobi (reference OceanCurrents.cpp:437-509) over latitude bands: pressure := 0, then
    `sweeps` sweeps in blocks of `ghost` with one halo refresh per block.  Returns the number of sweep launches.

    The refresh for block b+1 is posted as soon as the rows it carries exist -- after the edge tiles of block b's last
    launch -- and overlaps the interior tiles of that launch (dmg_ocean_jacobi_sweeps_begin/_end); DMG_HALO_OVERLAP=0
    (or an ocean without the split call) falls back to exchange-then-sweep.  Both orders give the same bits."""
    if zero_first:
        ocean.zero_pressure()
    if exch is None or exch.world == 1:
        ocean.jacobi_sweeps(sweeps)
        return sweeps
    overlap = _overlap_enabled(ocean)
    done = 0
    if overlap and sweeps > 0:
        exch.exchange(ocean, field, ghost)              # ghost rows for the first block
    while done < sweeps:
        n = min(ghost, sweeps - done)
        if not overlap:
            exch.exchange(ocean, field, ghost)
            ocean.jacobi_sweeps(n)
        elif done + n < sweeps:
            ocean.jacobi_sweeps_begin(n)                # ... up to the edge tiles of the block's last launch
            token = exch.start(ocean, field, ghost)     # their rows travel
            ocean.jacobi_sweeps_end()                   # while the interior tiles are swept
            exch.finish(token)
        else:
            ocean.jacobi_sweeps(n)                      # last block: nobody waits for its rows
        done += n
    return sweeps


F_PRESSURE, F_DIVW, F_TERRAIN, F_V = 0, 1, 2, 3     # DMG_F_* of include/demiurge_hip.h


def ocean_step(ocean, exch, n_outer=1):
    """OceanCurrents::run body (reference OceanCurrents.cpp:155-209) over latitude bands.

    Same passes in the same order as dmg_ocean_step; between them the ghost rows of the field the next pass reads
    are refreshed (depth = the ocean's `ghost`).  Bit-identical to the single-band step.
    """
    prm, g = ocean.params, ocean.ghost
    if exch is None or exch.world == 1:
        ocean.step(n_outer)
        return
    for _ in range(n_outer):
        exch.exchange(ocean, F_V, g)                    # advect gathers v up to advect_reach() rows away
        ocean.advect()                                  # :158
        done = 0
        while done < prm.diffusion_iters:               # :160, 50 sweeps in blocks of `ghost`
            n = min(g, prm.diffusion_iters - done)
            exch.exchange(ocean, F_V, g)
            ocean.diffusion(n)
            done += n
        exch.exchange(ocean, F_V, g)
        ocean.divergence()                              # :162  divw valid ghost-1 rows deep
        jacobi_solve(ocean, exch, prm.jacobi_iters, g - (g % 8) if g >= 8 else g, zero_first=True)   # :166
        exch.exchange(ocean, F_PRESSURE, g)
        ocean.subdiv()                                  # :168
