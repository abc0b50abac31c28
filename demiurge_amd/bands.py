"""Latitude-band driver over torch.distributed -- TEST AND BENCH TOOLING since round 2, not a product path.

The library drives its bands itself (demiurge_amd/csrc/multi.hip, dmg_multi: threads, streams, RCCL / peer-copy / direct-store
transports, failure handling, the deep advect halo).  This module is what round 1 used before that existed; it stays as (i) the
multi-process gloo tests' driver (tests/test_bands_gloo.py: CPU with the oracle, GPU with the library), (ii) an example of a
host that brings its own transport to dmg_ocean bands (dmg_ocean_halo), and (iii) the only place the pipelined edge chain
(jacobi_solve_pipelined, needs the tile kernels' row grid) lives -- an experiment that dmg_multi's edge-first order replaced.

Latitude-band driver: one process per GPU, pressure ghost rows refreshed with point-to-point exchanges.

The reference has no distributed piece; this is the only communication the backend
introduces.  A band owns H/n contiguous rows x full W, so the longitude wrap and the pole reflection stay
band-local.  Every `ghost` sweeps each rank sends its `ghost` outermost owned rows to the
neighbouring rank and receives that rank's rows into its ghost slab: the same bytes as a 1-row exchange per
sweep, in 1/ghost as many messages (xGMI is latency-bound at 64 KiB).  Results are bit-identical to one band.

Transport: `torch.distributed` P2P -- backend "nccl" (= RCCL over xGMI) sends straight from / into the library's
device slabs (zero copy); with backend "gloo" (CPU tests, or several ranks sharing one GPU) slabs are staged
through host tensors.  In the Jacobi solve the exchange for the next block of sweeps overlaps the interior tiles of the
current block's last launch when DMG_HALO_OVERLAP=1 (jacobi_solve).

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

    def start(self, ocean, field, rows, buffer=None):
        """Post the sends and receives that refresh `rows` ghost rows of `field` on both sides of this rank's band and
        return a token for finish().  With backend nccl the transfers run on RCCL's stream behind everything issued so
        far on torch's current stream; whatever the caller launches before finish() overlaps them.
        buffer: which of the two pressure buffers (default: the ocean's current one)."""
        dist, torch = self.dist, self.torch
        ops, post = [], []
        for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):     # DMG_SIDE_SOUTH, DMG_SIDE_NORTH
            if peer < 0 or peer >= self.world:
                continue
            send, recv, nbytes = ocean.halo(field, side, rows) if buffer is None else ocean.pressure_halo(buffer, side, rows)
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
    return os.environ.get("DMG_HALO_OVERLAP", "0") == "1" and hasattr(ocean, "jacobi_sweeps_begin")


def jacobi_solve(ocean, exch, sweeps, ghost, zero_first=True, field=0):
    """OceanCurrents::jacobi (reference OceanCurrents.cpp:437-509) over latitude bands: pressure := 0, then
    `sweeps` sweeps in blocks of `ghost` with one halo refresh per block.  Returns the number of sweep launches.

    Default order: exchange, then sweep.  Two overlapped orders are implemented and bit-identical, both opt-in because
    on one GPU (two bands exchanging through RCCL with the process's own rank, tools/nccl_self_probe.py) the plain order
    measured fastest (profiles/r01_rccl_self_probe.log) and there is no second GPU here to measure them across xGMI:
      DMG_HALO_OVERLAP=1   the refresh for block b+1 is posted after the edge tiles of block b's last launch and overlaps
                           the interior tiles of that launch (dmg_ocean_jacobi_sweeps_begin/_end);
      DMG_HALO_PIPELINE=1  jacobi_solve_pipelined: every launch split, the edge chain on a second stream."""
    import os
    if exch is not None and exch.world > 1 and os.environ.get("DMG_HALO_PIPELINE") == "1":
        return jacobi_solve_pipelined(ocean, exch, sweeps, ghost, zero_first)      # opt-in: see its docstring
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


def _block_plan(ocean, n, ghost):
    """Edge / interior split of the n/8 launches of a block: for launch k its rows [lo, hi), and the tile-row boundaries
    a <= b such that the tile rows below a and above b (its edge part) are all that the edge part of launch k+1 and the
    halo exchange need.  None when the block cannot be pipelined (tile shape still being tuned, odd sweep count)."""
    if n % 8 or n < 8:
        return None
    L, gs, gn = ocean.band_rows()
    m = n // 8
    geo = [ocean.jacobi_geometry(8 * (k + 1)) for k in range(m)]
    if any(g[2] == 0 for g in geo):
        return None
    need_s, need_n = gs + ghost, L - gn - ghost          # the rows to send: [gs, need_s) and [need_n, L - gn)
    plan = [None] * m
    for k in reversed(range(m)):
        lo, hi, oh = geo[k]
        a = min(hi, lo + oh * -(-(need_s - lo) // oh)) if gs else lo      # first tile-row boundary >= need_s
        b = max(lo, lo + oh * ((need_n - lo) // oh)) if gn else hi        # last tile-row boundary <= need_n
        plan[k] = (lo, hi, a, b)
        need_s, need_n = a + 8, b - 8                    # launch k reads 8 rows beyond what it writes
    return plan


def jacobi_solve_pipelined(ocean, exch, sweeps, ghost, zero_first=True):
    # `ocean` may be a list of bands living in this process (tools/nccl_self_probe.py: the exchanger then takes the list)
    """jacobi_solve with the halo exchange hidden behind the interior sweeps of a whole block.

    Every launch of a block is issued as two parts of one tiling (dmg_ocean_jacobi_launch): the tile rows next to the
    band edges on a second, high-priority stream, the rest on the compute stream.  The edge parts form a chain of their
    own -- the edge part of launch k+1 reads nothing but the edge part of launch k -- so that stream runs ahead, reaches
    the rows the neighbours wait for after about a third of the block's work, and posts the exchange there; the ghost
    rows arrive while the interior parts are still being swept.  Events order the two streams where they meet:
    interior k+1 after edge k, the next block's first edge part after this block's last interior part and the exchange.
    Same tiles, same arithmetic: bit-identical to jacobi_solve.  Falls back to it per block while a launch geometry's
    tile shape is still being tuned, for the last block, and altogether without two device streams (gloo, bounce).

    Opt-in (DMG_HALO_PIPELINE=1).  The two streams only run side by side when the HIP runtime may use more than its
    default four hardware queues (GPU_MAX_HW_QUEUES=8 in the environment before the process starts); with four, the
    edge and interior launches serialise and the block takes 1.5x as long.  Measured on one GPU with two bands
    exchanging through RCCL with the process's own rank (tools/nccl_self_probe.py, 8-rank band size, per 32-sweep
    block): no exchange 690 us; with 8 queues exchange-then-sweep 870 / pipelined 790 us; with the default 4 queues
    exchange-then-sweep 750 us.  What it does across xGMI, one band per GPU, is not measurable on one GPU."""
    torch = exch.torch
    oceans = list(ocean) if isinstance(ocean, (list, tuple)) else [ocean]
    if (exch.staged or exch.bounce or not all(hasattr(o, "jacobi_launch") for o in oceans) or not torch.cuda.is_available()):
        return jacobi_solve(ocean, exch, sweeps, ghost, zero_first)
    if zero_first:
        for o in oceans:
            o.zero_pressure()
    dev = exch.device if exch.device is not None else torch.device("cuda", torch.cuda.current_device())
    s_int = torch.cuda.current_stream(dev)               # the stream the ocean's context was created on (module docstring)
    if not hasattr(exch, "_edge_stream"):
        exch._edge_stream = torch.cuda.Stream(device=dev, priority=-1)
    s_edge = exch._edge_stream
    # invariant at the top of the loop: the ghost rows are fresh; `ev_ready` marks the interior stream's progress and
    # `ev_last_edge` (if any) the edge stream's, including the exchange it waited for
    if sweeps > 0:
        exch.exchange(ocean, F_PRESSURE, ghost)
    ev_ready = torch.cuda.Event()
    ev_ready.record(s_int)
    ev_last_edge = None
    done = 0
    while done < sweeps:
        n = min(ghost, sweeps - done)
        last = done + n >= sweeps
        plans = None if last else [_block_plan(o, n, ghost) for o in oceans]
        plan = None if plans is None or any(p is None for p in plans) else plans
        if plan is None:                                 # plain order: sweep, then refresh the ghost rows for the next block
            if ev_last_edge is not None:
                s_int.wait_event(ev_last_edge)
                ev_last_edge = None
            for o in oceans:
                o.jacobi_sweeps(n)
            if not last:
                exch.exchange(ocean, F_PRESSURE, ghost)
            ev_ready = torch.cuda.Event()
            ev_ready.record(s_int)
            done += n
            continue
        m = n // 8
        c = oceans[0].pressure_buffer()                  # every band of the process is at the same buffer
        s_edge.wait_event(ev_ready)                      # edge part 1 reads rows the previous interior part wrote
        if ev_last_edge is not None:
            s_int.wait_event(ev_last_edge)               # interior part 1 reads / overwrites rows of the previous edge chain
        ev_edge = []
        for k in range(m):
            for o, pl in zip(oceans, plan):
                lo, hi, a, b = pl[k]
                o.jacobi_launch(lo, hi, 8, 1, a, b, (c + k) & 1, s_edge.cuda_stream)
            e = torch.cuda.Event()
            e.record(s_edge)
            ev_edge.append(e)
        with torch.cuda.stream(s_edge):                  # RCCL orders itself behind the edge chain only
            token = exch.start(ocean, F_PRESSURE, ghost, buffer=(c + m) & 1)
        for k in range(m):
            if k:
                s_int.wait_event(ev_edge[k - 1])         # interior part k reads up to 8 rows of edge part k-1
            for o, pl in zip(oceans, plan):
                lo, hi, a, b = pl[k]
                o.jacobi_launch(lo, hi, 8, 2, a, b, (c + k) & 1, None)
        ev_ready = torch.cuda.Event()
        ev_ready.record(s_int)
        with torch.cuda.stream(s_edge):
            exch.finish(token)                           # the edge stream waits for the ghost rows
        ev_last_edge = torch.cuda.Event()
        ev_last_edge.record(s_edge)
        for o in oceans:
            o.set_pressure_buffer((c + m) & 1, m, n)
        done += n
    if ev_last_edge is not None:
        s_int.wait_event(ev_last_edge)
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
        # the back-trace of advect travels |v| * timestep: every rank learns the longest reach BEFORE anybody moves, so
        # that all of them go on or raise together (one rank failing alone would leave its peers in the next exchange)
        reach = ocean.advect_reach()
        t = exch.torch.tensor([reach], dtype=exch.torch.int32, device=exch.device if not exch.staged else "cpu")
        exch.dist.all_reduce(t, op=exch.dist.ReduceOp.MAX)
        reach = int(t[0])
        if reach > g:
            raise ValueError("advect reaches %d rows but the bands keep %d ghost rows: create the oceans with ghost >= %d "
                             "(every rank stopped before the step)" % (reach, g, reach))
        exch.exchange(ocean, F_V, g)                    # advect gathers v up to `reach` rows away
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
