// dmg_multi: one ocean split into latitude bands over several GPUs, driven by the library itself.
//
// SURVEY.md section 8b/8e asked for the backend to own its multi-GPU piece: one stream (and one host thread) per GPU,
// RCCL send/recv of the halo slabs over xGMI, one all-reduce for the advect reach.  The seam it replaces is the
// reference's single worker thread per filter (src/filter/Filter.cpp:125-166): an AsyncSubFilter's run() calls
// dmg_multi_* exactly as it would call dmg_ocean_*, and the bands, their streams, their worker threads and every
// exchange live below the C ABI.
//
// Two ways in:
//   dmg_multi_create        one process drives n bands (device_ids[i] = the GPU of band i).  Bands on distinct GPUs
//                           exchange through RCCL (ncclCommInitAll, grouped ncclSend/ncclRecv on each band's stream) or
//                           through peer copies (hipMemcpyPeerAsync + events); several bands on ONE GPU (tests, rehearsal)
//                           use the copies, or -- to run the RCCL calls themselves on a one-GPU box -- a loopback
//                           communicator of size 1 that sends every slab to the process's own rank.
//   dmg_multi_create_rank   one process per GPU (torchrun, MPI): each rank owns one band and a ncclCommInitRank
//                           communicator built from a 128-byte id that the host distributes.
// Results are bit-identical to one band whatever the split and the transport: the bands run the same kernels on the same
// rows, and a ghost row is a copy of the neighbour's row.
#include "ocean_internal.h"
#include <rccl/rccl.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <chrono>
#include <mutex>
#include <thread>

namespace {

#define DMG_NCCL(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess)                                                                          \
            return dmg_fail(DMG_EHIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

// the band threads of one process: run a closure on every band, in parallel, and meet at barriers inside it
struct band_team {
    int n = 0;
    std::vector<std::thread> threads;
    std::mutex mtx;
    std::condition_variable cv_go, cv_done, cv_bar;
    std::function<int(int)> job;
    long long generation = 0;
    int remaining = 0;
    bool quit = false;
    std::vector<int> rc;
    std::vector<std::string> err;
    // barrier among the n band threads (or a no-op for the caller's thread when n == 1)
    int bar_count = 0;
    long long bar_gen = 0;
    bool failed = false;             // a band has returned an error in this run(): barriers let everybody go
    std::function<void(int)> on_fail;    // called once by the band that failed, before the others are let go

    void start(int n_)
    {
        n = n_;
        rc.assign(n, 0);
        err.assign(n, std::string());
        if (n <= 1) return;
        for (int i = 0; i < n; i++)
            threads.emplace_back([this, i] {
                long long seen = 0;
                for (;;) {
                    std::function<int(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(mtx);
                        cv_go.wait(lk, [&] { return quit || generation != seen; });
                        if (quit) return;
                        seen = generation;
                        f = job;
                    }
                    int r = f(i);
                    std::string e = r ? dmg_last_error() : "";
                    if ((r == DMG_EHIP || r == DMG_ENOMEM) && on_fail) on_fail(i);    // (a refused argument is collective and leaves nothing queued)
                    {
                        std::lock_guard<std::mutex> lk(mtx);
                        rc[i] = r;
                        err[i] = e;
                        if (r) { failed = true; cv_bar.notify_all(); }     // nobody waits for a band that has given up
                        if (--remaining == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mtx);
            quit = true;
        }
        cv_go.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
    }
    // runs f(i) for every band; returns the first failure (message copied into the caller's dmg_last_error)
    int run(const std::function<int(int)> &f)
    {
        if (n <= 1) {
            const int r = n == 1 ? f(0) : DMG_OK;
            if ((r == DMG_EHIP || r == DMG_ENOMEM) && on_fail) { const std::string keep = dmg_last_error(); on_fail(0); return dmg_fail(r, "%s", keep.c_str()); }
            return r;
        }
        {
            std::lock_guard<std::mutex> lk(mtx);
            job = f;
            remaining = n;
            failed = false;
            bar_count = 0;
            generation++;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(mtx);
        cv_done.wait(lk, [&] { return remaining == 0; });
        for (int i = 0; i < n; i++)                          // the band that failed first, not one that was merely let go
            if (rc[i] && rc[i] != DMG_ESTATE) return dmg_fail(rc[i], "band %d: %s", i, err[i].c_str());
        for (int i = 0; i < n; i++)
            if (rc[i]) return dmg_fail(rc[i], "band %d: %s", i, err[i].c_str());
        return DMG_OK;
    }
    // false: another band has failed, the caller should return too
    bool barrier()
    {
        if (n <= 1) return true;
        std::unique_lock<std::mutex> lk(mtx);
        if (failed) return false;
        const long long g = bar_gen;
        if (++bar_count == n) { bar_count = 0; bar_gen++; cv_bar.notify_all(); }
        else cv_bar.wait(lk, [&] { return failed || bar_gen != g; });
        return !failed;
    }
};

}  // namespace

struct dmg_multi {
    int world = 1;                   // bands in the whole job
    int first = 0;                   // global index of this process's first band
    int n = 0;                       // bands in this process
    int transport = DMG_MULTI_PEER;
    int order = DMG_MULTI_EXCHANGE_THEN_SWEEP;
    int ghost = 1;
    bool shared_ctx = false;         // loopback: every band on ctx[0] (one stream)
    dmg_grid grid{};
    dmg_ocean_params prm{};
    std::vector<dmg_ctx *> ctx;
    std::vector<dmg_ocean *> oc;     // the bands of an ocean (dmg_multi_create*), or ...
    std::vector<dmg_climate *> cl;   // ... the bands of a climate field (dmg_multi_climate_create*): same streams, threads and transports
    std::vector<ncclComm_t> comm;    // RCCL: one per local band (loopback: one in all)
    std::vector<hipEvent_t> ev_done, ev_copied;    // PEER: per band
    // DMG_MULTI_OVERLAP_LAST_LAUNCH: a second, high-priority stream per band carries the exchange while the band's own
    // stream sweeps the interior units of the block's last launch
    std::vector<hipStream_t> comm_stream;
    std::vector<hipEvent_t> ev_edge, ev_xchg;      // per band: a whole launch done (own stream) / ghost rows refreshed (second stream)
    std::atomic<bool> dead{false};   // a band failed and the communicators were aborted: only dmg_multi_destroy remains
    int timeout_ms = 120000;         // RCCL transports: how long a host-side wait may take before the communicators are aborted
    int fake_link_us = 0;            // experiments (DMG_MULTI_FAKE_LINK_US=<us>): every exchange is followed, on its stream, by a kernel that
                                     // waits that long -- a link slower than a local copy, to see what the overlapped order hides of it
    int fail_band = -1;              // tests (DMG_MULTI_FAIL=<band>:<n>): that band gives up in front of its n-th exchange
    long long fail_at = 0;
    std::vector<long long> n_band_exchanges;
    std::vector<int *> d_red;        // per band: one device word for reductions
    band_team team;
    // halo-exchange time: event pairs on band 0's stream around every exchange, folded into the sum when read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free, ev_used;
    double exchange_ms = 0.0;
    long long n_exchanges = 0, exchange_bytes = 0;
    dmg_ctx *ctx_of(int i) const { return shared_ctx ? ctx[0] : ctx[i]; }
};

static bool multi_trace() { static const bool on = getenv("DMG_MULTI_TRACE") != nullptr; return on; }
#define MTRACE(...) do { if (multi_trace()) { fprintf(stderr, "[multi] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

#define TEAM_BARRIER(m) do { if (!(m)->team.barrier()) return dmg_fail(DMG_ESTATE, "stopped: another band failed"); } while (0)

static int multi_fold_events(dmg_multi *m)
{
    for (auto &p : m->ev_used) {
        float ms = 0.0f;
        DMG_HIP(hipEventSynchronize(p.second));
        DMG_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        m->exchange_ms += ms;
        m->ev_free.push_back(p);
    }
    m->ev_used.clear();
    return DMG_OK;
}

// A band has failed, or a wait has run out: abort the local communicators so that receives still queued on the streams end
// instead of waiting for a peer that will never send (ncclCommAbort tears the communicator's kernels down), and mark the
// handle dead.  The bands of other processes notice through their own timed waits (multi_wait).
static void multi_abort(dmg_multi *m)
{
    bool was = false;
    if (!m->dead.compare_exchange_strong(was, true)) return;
    for (ncclComm_t &c : m->comm)
        if (c) { (void)ncclCommAbort(c); c = nullptr; }
}

// wait for band i's streams.  With RCCL a peer that has died leaves a receive kernel spinning forever, so the wait polls,
// watches the communicator's asynchronous error state, and gives up after timeout_ms (DMG_MULTI_TIMEOUT_MS): it aborts
// the communicators and returns an error instead of hanging the caller.
static int multi_wait(dmg_multi *m, int i)
{
    dmg_ctx *c = m->ctx_of(i);
    DMG_HIP(hipSetDevice(c->device));
    const bool rccl = (m->transport == DMG_MULTI_RCCL || m->transport == DMG_MULTI_RCCL_LOOPBACK) && !m->comm.empty();
    if (!rccl || m->dead.load()) { DMG_HIP(hipStreamSynchronize(c->stream)); return DMG_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    ncclComm_t comm = m->comm[m->transport == DMG_MULTI_RCCL_LOOPBACK ? 0 : i];
    for (long long spin = 0;; spin++) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return DMG_OK;
        if (q != hipErrorNotReady) return dmg_fail(DMG_EHIP, "band %d: stream failed: %s", m->first + i, hipGetErrorString(q));
        if ((spin & 63) == 63) {
            ncclResult_t async = ncclSuccess;
            if (comm && ncclCommGetAsyncError(comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                multi_abort(m);
                return dmg_fail(DMG_EHIP, "band %d: RCCL reports %s; communicators aborted", m->first + i, ncclGetErrorString(async));
            }
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > m->timeout_ms) {
                multi_abort(m);
                return dmg_fail(DMG_EHIP, "band %d: no progress for %d ms (a neighbour band has probably failed); communicators aborted "
                                          "(DMG_MULTI_TIMEOUT_MS)", m->first + i, m->timeout_ms);
            }
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
        else std::this_thread::yield();
    }
}

// ---- one halo exchange of `field` (rows deep) as seen by local band i; every band thread calls it ----
// `on`: the stream that carries it (default: the band's own)
// experiments: one wave that holds its stream for `ticks` of the 100 MHz clock (DMG_MULTI_FAKE_LINK_US)
__global__ void k_multi_fake_link(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static int band_exchange(dmg_multi *m, int i, int field, int rows, int buffer = -1, hipStream_t on = nullptr)
{
    dmg_ctx *c = m->ctx_of(i);
    const int g = m->first + i;
    MTRACE("band %d: exchange field %d rows %d transport %d", g, field, rows, m->transport);
    if (m->dead.load()) return dmg_fail(DMG_ESTATE, "this dmg_multi has lost a band: destroy it");
    if (g == m->fail_band && ++m->n_band_exchanges[i] == m->fail_at)
        return dmg_fail(DMG_EHIP, "band %d: failure injected in front of exchange %lld (DMG_MULTI_FAIL)", g, m->fail_at);
    DMG_HIP(hipSetDevice(c->device));
    hipStream_t st = on ? on : c->stream;
    std::pair<hipEvent_t, hipEvent_t> ev{ nullptr, nullptr };
    if (i == 0) {                                            // band 0 times the exchange on its stream
        if (m->ev_used.size() >= 2048) { int rc = multi_fold_events(m); if (rc) return rc; }
        if (m->ev_free.empty()) {
            DMG_HIP(hipEventCreate(&ev.first));
            DMG_HIP(hipEventCreate(&ev.second));
        } else { ev = m->ev_free.back(); m->ev_free.pop_back(); }
        DMG_HIP(hipEventRecord(ev.first, st));
    }
    auto slab = [&](int band, int side, void **send, void **recv, size_t *bytes) {
        if (!m->cl.empty()) return dmg_climate_halo(m->cl[band], side, rows, send, recv, bytes);      // the temperature field's ghost rows
        dmg_ocean *oo = m->oc[band];
        return (field == DMG_F_PRESSURE && buffer >= 0) ? dmg_ocean_pressure_halo(oo, buffer, side, rows, send, recv, bytes)
                                                        : dmg_ocean_halo(oo, field, side, rows, send, recv, bytes);
    };
    int rc;
    if (m->transport == DMG_MULTI_RCCL) {
        // grouped sends and receives on this band's stream; RCCL pairs them with the neighbour's calls.  Whatever happens
        // inside the group, the group is closed: a thread left in group mode would have its later RCCL calls deferred
        DMG_NCCL(ncclGroupStart());
        ncclResult_t first = ncclSuccess;
        rc = DMG_OK;
        for (int side = 0; side < 2 && !rc && first == ncclSuccess; side++) {
            const int peer = side == DMG_SIDE_SOUTH ? g - 1 : g + 1;
            if (peer < 0 || peer >= m->world) continue;
            void *send, *recv; size_t bytes;
            if ((rc = slab(i, side, &send, &recv, &bytes))) break;
            first = ncclSend(send, bytes / 4, ncclFloat, peer, m->comm[i], st);
            if (first == ncclSuccess) first = ncclRecv(recv, bytes / 4, ncclFloat, peer, m->comm[i], st);
            if (i == 0) m->exchange_bytes += 2 * (long long)bytes;
        }
        const ncclResult_t closed = ncclGroupEnd();
        if (rc) return rc;
        if (first != ncclSuccess || closed != ncclSuccess)
            return dmg_fail(DMG_EHIP, "band %d: RCCL send/recv failed: %s", g, ncclGetErrorString(first != ncclSuccess ? first : closed));
    } else if (m->transport == DMG_MULTI_RCCL_LOOPBACK) {
        // every band on one stream, one communicator of size 1: once every band's thread has enqueued its block, band 0's
        // thread walks the boundaries and sends each slab to the process's own rank (send and receive of a pair are
        // matched in issue order); nobody enqueues the next block before that
        TEAM_BARRIER(m);
        if (i == 0) {
            DMG_NCCL(ncclGroupStart());
            ncclResult_t first = ncclSuccess;
            rc = DMG_OK;
            for (int b = 0; b + 1 < m->n && !rc && first == ncclSuccess; b++) {
                void *s_lo, *r_lo, *s_hi, *r_hi; size_t by;
                if ((rc = slab(b, DMG_SIDE_NORTH, &s_lo, &r_lo, &by)) || (rc = slab(b + 1, DMG_SIDE_SOUTH, &s_hi, &r_hi, &by))) break;
                first = ncclSend(s_lo, by / 4, ncclFloat, 0, m->comm[0], st);                               // band b's top rows ...
                if (first == ncclSuccess) first = ncclRecv(r_hi, by / 4, ncclFloat, 0, m->comm[0], st);     // ... into band b+1's south ghost
                if (first == ncclSuccess) first = ncclSend(s_hi, by / 4, ncclFloat, 0, m->comm[0], st);
                if (first == ncclSuccess) first = ncclRecv(r_lo, by / 4, ncclFloat, 0, m->comm[0], st);
                m->exchange_bytes += 2 * (long long)by;
            }
            const ncclResult_t closed = ncclGroupEnd();
            if (rc) return rc;
            if (first != ncclSuccess || closed != ncclSuccess)
                return dmg_fail(DMG_EHIP, "RCCL loopback send/recv failed: %s", ncclGetErrorString(first != ncclSuccess ? first : closed));
        }
        TEAM_BARRIER(m);
    } else {
        // peer copies: I pull my ghost rows from the neighbours' owned rows on MY stream, once they have finished the
        // block that produces them; they continue once I have read their rows
        DMG_HIP(hipEventRecord(m->ev_done[i], st));
        TEAM_BARRIER(m);                                     // every band's event is recorded
        for (int side = 0; side < 2; side++) {
            const int j = side == DMG_SIDE_SOUTH ? i - 1 : i + 1;
            if (j < 0 || j >= m->n) continue;
            void *send, *recv, *nsend, *nrecv; size_t bytes, nbytes;
            if ((rc = slab(i, side, &send, &recv, &bytes)) || (rc = slab(j, side ^ 1, &nsend, &nrecv, &nbytes))) return rc;
            if (m->ctx_of(j) != c) DMG_HIP(hipStreamWaitEvent(st, m->ev_done[j], 0));
            DMG_HIP(hipMemcpyPeerAsync(recv, c->device, nsend, m->ctx_of(j)->device, bytes, st));
            if (i == 0) m->exchange_bytes += (long long)bytes;
        }
        DMG_HIP(hipEventRecord(m->ev_copied[i], st));
        TEAM_BARRIER(m);
        for (int j : { i - 1, i + 1 })                       // the neighbours have read my rows before I overwrite them
            if (j >= 0 && j < m->n && m->ctx_of(j) != c) DMG_HIP(hipStreamWaitEvent(st, m->ev_copied[j], 0));
    }
    if (m->fake_link_us > 0 && (m->transport != DMG_MULTI_RCCL_LOOPBACK || i == 0))
        hipLaunchKernelGGL(k_multi_fake_link, dim3(1), dim3(64), 0, st, (unsigned long long)m->fake_link_us * 100ull);
    if (i == 0) {
        DMG_HIP(hipEventRecord(ev.second, st));
        m->ev_used.push_back(ev);
        m->n_exchanges++;
    }
    MTRACE("band %d: exchange done", g);
    return DMG_OK;
}

// A block whose last launch is split: the edge units of that launch and the exchange behind them on the band's second
// stream, the interior units on the band's own stream at the same time; the own stream waits for the exchange in front of
// the next block.
static int band_block_overlapped(dmg_multi *m, int i, int field, int rows, int n)
{
    dmg_ocean *o = m->oc[i];
    dmg_ctx *c = m->ctx_of(i);
    int rc;
    DMG_HIP(hipSetDevice(c->device));
    hipStream_t cs = m->comm_stream[m->shared_ctx ? 0 : i];
    if ((rc = ocean_jacobi_sweeps_begin(o, n, cs))) return rc;             // launches 1 .. m-1 (own stream), edge units of launch m (second stream)
    if (!o->pending.active) {
        // the launch was issued whole on the own stream (its shape is still being tuned, or it has no blocked form): the
        // exchange waits for it
        DMG_HIP(hipEventRecord(m->ev_edge[i], c->stream));
        DMG_HIP(hipStreamWaitEvent(cs, m->ev_edge[i], 0));
    }
    if (!m->shared_ctx) {
        if ((rc = band_exchange(m, i, field, rows, -1, cs))) return rc;   // behind the edge units
        DMG_HIP(hipEventRecord(m->ev_xchg[i], cs));
        if ((rc = dmg_ocean_jacobi_sweeps_end(o))) return rc;             // interior units: own stream, meanwhile
        DMG_HIP(hipStreamWaitEvent(c->stream, m->ev_xchg[i], 0));         // the next block reads the refreshed ghost rows
        return DMG_OK;
    }
    // loopback: every band on one pair of streams.  band_exchange (band 0's thread, between two barriers) queues the whole
    // exchange behind every band's edge units; band 0 orders the streams once everybody has queued its interior units
    if ((rc = band_exchange(m, i, field, rows, -1, cs))) return rc;
    if (i == 0) DMG_HIP(hipEventRecord(m->ev_xchg[0], cs));
    if ((rc = dmg_ocean_jacobi_sweeps_end(o))) return rc;
    TEAM_BARRIER(m);                                                       // every band's interior units are queued
    if (i == 0) DMG_HIP(hipStreamWaitEvent(c->stream, m->ev_xchg[0], 0));
    TEAM_BARRIER(m);                                                       // nobody queues the next block in front of that wait
    return DMG_OK;
}

// max over all bands of a per-band integer (the advect reach): one ncclAllReduce, or the host when one process holds all
static int band_allreduce_max(dmg_multi *m, int i, int *value, std::vector<int> &scratch)
{
    dmg_ctx *c = m->ctx_of(i);
    if (m->transport == DMG_MULTI_RCCL) {
        DMG_HIP(hipSetDevice(c->device));
        DMG_HIP(hipMemcpyAsync(m->d_red[i], value, sizeof(int), hipMemcpyHostToDevice, c->stream));
        DMG_NCCL(ncclAllReduce(m->d_red[i], m->d_red[i], 1, ncclInt, ncclMax, m->comm[i], c->stream));
        DMG_HIP(hipMemcpyAsync(value, m->d_red[i], sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return multi_wait(m, i);
    }
    scratch[i] = *value;                                     // one process: the bands meet on the host
    TEAM_BARRIER(m);
    int mx = 0;
    for (int k = 0; k < m->n; k++) mx = std::max(mx, scratch[k]);
    TEAM_BARRIER(m);
    *value = mx;
    return DMG_OK;
}

static int band_jacobi(dmg_multi *m, int i, int iters, int zero_first)
{
    dmg_ocean *o = m->oc[i];
    int rc;
    if (zero_first && (rc = dmg_ocean_zero_pressure(o))) return rc;
    if (m->world == 1) return dmg_ocean_jacobi_sweeps(o, iters);
    const int block = m->ghost >= 8 ? m->ghost - (m->ghost % 8) : m->ghost;      // whole 8-sweep launches per block
    const bool overlap = m->order == DMG_MULTI_OVERLAP_LAST_LAUNCH;
    int done = 0;
    MTRACE("band %d: jacobi %d sweeps, block %d, overlap %d", m->first + i, iters, block, (int)overlap);
    if (overlap && iters > 0 && (rc = band_exchange(m, i, DMG_F_PRESSURE, m->ghost))) return rc;
    while (done < iters) {
        const int n = std::min(block, iters - done);
        if (!overlap) {
            if ((rc = band_exchange(m, i, DMG_F_PRESSURE, m->ghost))) return rc;
            if ((rc = dmg_ocean_jacobi_sweeps(o, n))) return rc;
        } else if (done + n < iters) {
            // the rows the neighbours wait for exist once the edge units of the block's last launch have run: those units
            // and the exchange behind them go to the band's second, high-priority stream, the interior units run on the
            // band's own stream at the same time -- they read the previous iterate and write neither the rows in flight
            // nor the ghost rows being filled -- and the own stream waits for the exchange only in front of the next
            // block.  (Filter.cpp:125-166: still one worker per band.)
            if ((rc = band_block_overlapped(m, i, DMG_F_PRESSURE, m->ghost, n))) return rc;
        } else if ((rc = dmg_ocean_jacobi_sweeps(o, n))) return rc;
        done += n;
    }
    return DMG_OK;
}

// OceanCurrents::run body (reference OceanCurrents.cpp:155-209) on one band; ghost rows of the field a pass reads are
// refreshed in front of it
static int band_step(dmg_multi *m, int i, int n_outer, std::vector<int> &scratch)
{
    dmg_ocean *o = m->oc[i];
    const int g = m->ghost;
    int rc;
    if (m->world == 1) return dmg_ocean_step(o, n_outer);
    for (int k = 0; k < n_outer; k++) {
        // the back-trace of advect travels |v| * timestep (:298-299): every band learns the longest reach before anybody
        // moves, so that all of them either go on or fail together, with v untouched
        int reach = 0;
        if ((rc = dmg_ocean_advect_reach(o, &reach))) return rc;
        if ((rc = band_allreduce_max(m, i, &reach, scratch))) return rc;
        if (reach > g)
            return dmg_fail(DMG_EINVAL, "advect reaches %d rows but the bands keep %d ghost rows: create the dmg_multi with "
                                        "ghost >= %d (every band stopped before the step, nothing was changed)", reach, g, reach);
        if ((rc = band_exchange(m, i, DMG_F_V, g))) return rc;
        if ((rc = dmg_ocean_advect(o))) return rc;                                   // :158
        for (int done = 0; done < m->prm.diffusion_iters;) {                          // :160, in blocks of `ghost` sweeps
            const int n = std::min(g, m->prm.diffusion_iters - done);
            if ((rc = band_exchange(m, i, DMG_F_V, g))) return rc;
            if ((rc = dmg_ocean_diffusion(o, n))) return rc;
            done += n;
        }
        if ((rc = band_exchange(m, i, DMG_F_V, g))) return rc;
        if ((rc = dmg_ocean_divergence(o))) return rc;                               // :162
        if ((rc = band_jacobi(m, i, m->prm.jacobi_iters, 1))) return rc;             // :166
        if ((rc = band_exchange(m, i, DMG_F_PRESSURE, g))) return rc;
        if ((rc = dmg_ocean_subdiv(o))) return rc;                                   // :168
        o->n_outer++;
    }
    return DMG_OK;
}

static void multi_free(dmg_multi *m)
{
    if (!m) return;
    m->team.stop();
    for (auto &p : m->ev_used) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : m->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (size_t i = 0; i < m->d_red.size(); i++)
        if (m->d_red[i] && m->ctx_of((int)i)) { (void)hipSetDevice(m->ctx_of((int)i)->device); (void)hipFree(m->d_red[i]); }
    for (dmg_ocean *o : m->oc) if (o) dmg_ocean_destroy(o);
    for (dmg_climate *k : m->cl) if (k) dmg_climate_destroy(k);
    for (hipEvent_t e : m->ev_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_copied) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_edge) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_xchg) if (e) (void)hipEventDestroy(e);
    for (size_t i = 0; i < m->comm_stream.size(); i++)
        if (m->comm_stream[i]) { (void)hipSetDevice(m->ctx_of((int)i)->device); (void)hipStreamSynchronize(m->comm_stream[i]); (void)hipStreamDestroy(m->comm_stream[i]); }
    for (ncclComm_t c : m->comm) if (c) (void)ncclCommDestroy(c);
    for (dmg_ctx *c : m->ctx) if (c) dmg_destroy(c);
    delete m;
}

static int multi_common(dmg_multi *m, const dmg_grid *grid, const dmg_ocean_params *params, int ghost, int flags)
{
    m->grid = *grid;
    if (params) m->prm = *params; else dmg_ocean_params_default(&m->prm);
    m->ghost = ghost;
    m->order = (flags & DMG_MULTI_OVERLAP_LAST_LAUNCH) ? DMG_MULTI_OVERLAP_LAST_LAUNCH : DMG_MULTI_EXCHANGE_THEN_SWEEP;
    if (const char *e = getenv("DMG_MULTI_TIMEOUT_MS")) { const int v = atoi(e); if (v > 0) m->timeout_ms = v; }
    if (const char *e = getenv("DMG_MULTI_FAIL")) {          // tests: "<band>:<n>"
        int b = -1; long long k = 0;
        if (sscanf(e, "%d:%lld", &b, &k) == 2 && b >= 0 && k > 0) { m->fail_band = b; m->fail_at = k; }
    }
    if (const char *e = getenv("DMG_MULTI_FAKE_LINK_US")) {
        const int us = atoi(e);
        if (us > 0 && us <= 100000) m->fake_link_us = us;
    }
    return DMG_OK;
}

// the second stream of every local band (one in all for the loopback transport) and its events; the failure hook
static int multi_streams(dmg_multi *m)
{
    const int ns = m->shared_ctx ? 1 : m->n;
    m->comm_stream.assign((size_t)ns, nullptr); m->ev_edge.assign((size_t)m->n, nullptr); m->ev_xchg.assign((size_t)m->n, nullptr);
    m->n_band_exchanges.assign((size_t)m->n, 0);
    for (int i = 0; i < m->n; i++) {
        DMG_HIP(hipSetDevice(m->ctx_of(i)->device));
        if (i < ns) {
            int lo = 0, hi = 0;
            DMG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (hi = greatest priority, numerically lowest)
            static const bool normal = [] { const char *e = getenv("DMG_MULTI_COMM_PRIORITY"); return e && atoi(e) == 0; }();   // experiments
            DMG_HIP(hipStreamCreateWithPriority(&m->comm_stream[i], hipStreamNonBlocking, normal ? lo : hi));
        }
        // experiments (DMG_MULTI_EVENT_FLAGS=2): the join event with an agent-scope release too.  The default keeps the
        // system-scope release: what it orders was written by a neighbour GPU or a copy engine
        static const int evf = [] { const char *e = getenv("DMG_MULTI_EVENT_FLAGS"); return e ? atoi(e) : 1; }();
        DMG_HIP(hipEventCreateWithFlags(&m->ev_edge[i], hipEventDisableTiming));
        DMG_HIP(hipEventCreateWithFlags(&m->ev_xchg[i], hipEventDisableTiming | (evf >= 2 ? hipEventReleaseToDevice : 0)));
    }
    m->team.on_fail = [m](int) { multi_abort(m); };
    return DMG_OK;
}

extern "C" {

int dmg_multi_unique_id(void *id128)
{
    DMG_CHECK_ARG(id128 != nullptr);
    static_assert(sizeof(ncclUniqueId) == DMG_MULTI_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    DMG_NCCL(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return DMG_OK;
}

// make_band(m, local index, global band index, bands in all): creates that band's object on m->ctx_of(local index)
typedef std::function<int(dmg_multi *, int, int, int)> band_factory;

static int multi_create_local(int n_bands, const int *device_ids, const dmg_grid *grid, const dmg_ocean_params *params, int ghost, int flags,
                              const band_factory &make_band, dmg_multi **out)
{
    DMG_CHECK_ARG(out != nullptr && n_bands >= 1 && n_bands <= 64 && device_ids != nullptr && grid != nullptr && ghost >= 1);
    *out = nullptr;
    bool distinct = true;
    for (int i = 0; i < n_bands; i++)
        for (int j = 0; j < i; j++) distinct = distinct && device_ids[i] != device_ids[j];
    int transport = flags & DMG_MULTI_TRANSPORT_MASK;
    if (transport == DMG_MULTI_AUTO) transport = (distinct && n_bands > 1) ? DMG_MULTI_RCCL : DMG_MULTI_PEER;
    if (transport == DMG_MULTI_RCCL && !distinct)
        return dmg_fail(DMG_EINVAL, "DMG_MULTI_RCCL needs one GPU per band (RCCL refuses two ranks on one device); use "
                                    "DMG_MULTI_PEER or DMG_MULTI_RCCL_LOOPBACK for several bands on one GPU");
    if (transport == DMG_MULTI_RCCL_LOOPBACK)
        for (int i = 1; i < n_bands; i++) DMG_CHECK_ARG(device_ids[i] == device_ids[0]);
    std::unique_ptr<dmg_multi, void (*)(dmg_multi *)> m(new (std::nothrow) dmg_multi, multi_free);
    if (!m) return dmg_fail(DMG_ENOMEM, "out of host memory");
    int rc = multi_common(m.get(), grid, params, ghost, flags);
    if (rc) return rc;
    m->world = n_bands; m->first = 0; m->n = n_bands; m->transport = transport;
    m->shared_ctx = transport == DMG_MULTI_RCCL_LOOPBACK;
    m->ctx.assign(n_bands, nullptr); m->d_red.assign(n_bands, nullptr);
    m->ev_done.assign(n_bands, nullptr); m->ev_copied.assign(n_bands, nullptr);
    for (int i = 0; i < n_bands; i++) {
        if (!m->shared_ctx || i == 0)
            if ((rc = dmg_create(device_ids[i], nullptr, DMG_PRIVATE_STREAM, &m->ctx[i]))) return rc;
        if ((rc = make_band(m.get(), i, i, n_bands))) return rc;
        DMG_HIP(hipSetDevice(device_ids[i]));
        DMG_HIP(hipMalloc((void **)&m->d_red[i], sizeof(int)));
        DMG_HIP(hipEventCreateWithFlags(&m->ev_done[i], hipEventDisableTiming));
        DMG_HIP(hipEventCreateWithFlags(&m->ev_copied[i], hipEventDisableTiming));
    }
    if (transport == DMG_MULTI_PEER)                         // let the copy engines reach the neighbour's memory directly
        for (int i = 0; i + 1 < n_bands; i++) {
            const int a = device_ids[i], b = device_ids[i + 1];
            if (a == b) continue;
            int ok = 0;
            (void)hipDeviceCanAccessPeer(&ok, a, b);
            if (ok) {
                (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0);
                (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0);
                (void)hipGetLastError();                     // "already enabled" is fine
            }
        }
    if (transport == DMG_MULTI_RCCL) {
        m->comm.assign(n_bands, nullptr);
        DMG_NCCL(ncclCommInitAll(m->comm.data(), n_bands, device_ids));
    } else if (transport == DMG_MULTI_RCCL_LOOPBACK) {
        m->comm.assign(1, nullptr);
        DMG_HIP(hipSetDevice(device_ids[0]));
        DMG_NCCL(ncclCommInitAll(m->comm.data(), 1, device_ids));
    }
    if ((rc = multi_streams(m.get()))) return rc;
    // one host thread per band, like the reference's one worker per filter
    m->team.start(n_bands);
    *out = m.release();
    return DMG_OK;
}

static int multi_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const dmg_ocean_params *params,
                             int ghost, int flags, const band_factory &make_band, dmg_multi **out)
{
    DMG_CHECK_ARG(out != nullptr && n_ranks >= 1 && rank >= 0 && rank < n_ranks && grid != nullptr && ghost >= 1);
    DMG_CHECK_ARG(n_ranks == 1 || id128 != nullptr);
    *out = nullptr;
    std::unique_ptr<dmg_multi, void (*)(dmg_multi *)> m(new (std::nothrow) dmg_multi, multi_free);
    if (!m) return dmg_fail(DMG_ENOMEM, "out of host memory");
    int rc = multi_common(m.get(), grid, params, ghost, flags);
    if (rc) return rc;
    m->world = n_ranks; m->first = rank; m->n = 1; m->transport = DMG_MULTI_RCCL;
    m->ctx.assign(1, nullptr); m->d_red.assign(1, nullptr);
    m->ev_done.assign(1, nullptr); m->ev_copied.assign(1, nullptr);
    if ((rc = dmg_create(device_id, nullptr, DMG_PRIVATE_STREAM, &m->ctx[0]))) return rc;
    if ((rc = make_band(m.get(), 0, rank, n_ranks))) return rc;
    DMG_HIP(hipSetDevice(device_id));
    DMG_HIP(hipMalloc((void **)&m->d_red[0], sizeof(int)));
    if (n_ranks > 1) {
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        m->comm.assign(1, nullptr);
        DMG_NCCL(ncclCommInitRank(&m->comm[0], n_ranks, id, rank));
    }
    if ((rc = multi_streams(m.get()))) return rc;
    m->team.start(1);
    *out = m.release();
    return DMG_OK;
}

static band_factory ocean_band(const dmg_grid *grid, int ghost)
{
    return [grid, ghost](dmg_multi *m, int i, int g, int world) {
        if (m->oc.size() < (size_t)m->ctx.size()) m->oc.resize(m->ctx.size(), nullptr);
        return dmg_ocean_create(m->ctx_of(i), grid, &m->prm, g, world, ghost, &m->oc[i]);
    };
}

int dmg_multi_create(int n_bands, const int *device_ids, const dmg_grid *grid, const dmg_ocean_params *params, int ghost, int flags,
                     dmg_multi **out)
{
    return multi_create_local(n_bands, device_ids, grid, params, ghost, flags, ocean_band(grid, ghost), out);
}

int dmg_multi_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const dmg_ocean_params *params,
                          int ghost, int flags, dmg_multi **out)
{
    return multi_create_rank(rank, n_ranks, id128, device_id, grid, params, ghost, flags, ocean_band(grid, ghost), out);
}

// ---- the climate year cycle over latitude bands --------------------------------------------------
// Temperature::run (src/filter/Temperature.cpp:47-162) steps the field 500 000 times: of config 5's stages it is the one
// that can use eight GPUs.  The bands are dmg_climate_create_band handles (filters.hip: up to `ghost` steps between halo
// refreshes, the computed region shrinking a row per step); threads, streams, transports and failure handling are
// dmg_multi's -- the exchange is the very band_exchange of the ocean, on the temperature field's slabs.
static band_factory climate_band(const dmg_grid *grid, const float *T, const float *terrain, int ghost)
{
    return [grid, T, terrain, ghost](dmg_multi *m, int i, int g, int world) {
        if (m->cl.size() < (size_t)m->ctx.size()) m->cl.resize(m->ctx.size(), nullptr);
        return dmg_climate_create_band(m->ctx_of(i), grid, T, terrain, g, world, world > 1 ? ghost : 0, &m->cl[i]);
    };
}

int dmg_multi_climate_create(int n_bands, const int *device_ids, const dmg_grid *grid, const float *T, const float *terrain, int ghost, int flags,
                             dmg_multi_climate **out)
{
    DMG_CHECK_ARG(T != nullptr && terrain != nullptr && out != nullptr);
    dmg_multi *m = nullptr;
    int rc = multi_create_local(n_bands, device_ids, grid, nullptr, ghost, flags, climate_band(grid, T, terrain, ghost), &m);
    *out = reinterpret_cast<dmg_multi_climate *>(m);
    return rc;
}

int dmg_multi_climate_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const float *T, const float *terrain,
                                  int ghost, int flags, dmg_multi_climate **out)
{
    DMG_CHECK_ARG(T != nullptr && terrain != nullptr && out != nullptr);
    dmg_multi *m = nullptr;
    int rc = multi_create_rank(rank, n_ranks, id128, device_id, grid, nullptr, ghost, flags, climate_band(grid, T, terrain, ghost), &m);
    *out = reinterpret_cast<dmg_multi_climate *>(m);
    return rc;
}

void dmg_multi_climate_destroy(dmg_multi_climate *h) { multi_free(reinterpret_cast<dmg_multi *>(h)); }

// n_steps of the energy-balance model from step index first_step on (the orbital phase is 2 pi / 15000 * step, :146); the
// ghost rows are refreshed every `ghost` steps.  Asynchronous on the bands' streams.
int dmg_multi_climate_steps(dmg_multi_climate *h, int first_step, int n_steps)
{
    dmg_multi *m = reinterpret_cast<dmg_multi *>(h);
    DMG_CHECK_ARG(m != nullptr && !m->cl.empty() && n_steps >= 0);
    if (m->dead.load()) return dmg_fail(DMG_ESTATE, "this handle has lost a band: destroy it");
    return m->team.run([&](int i) -> int {
        dmg_climate *k = m->cl[i];
        if (m->world == 1) return dmg_climate_steps(k, first_step, n_steps);
        int rc;
        for (int done = 0; done < n_steps;) {
            const int n = std::min(m->ghost, n_steps - done);
            if ((rc = band_exchange(m, i, 0, m->ghost))) return rc;
            if ((rc = dmg_climate_steps(k, first_step + done, n))) return rc;
            done += n;
        }
        return DMG_OK;
    });
}

int dmg_multi_climate_set_terrain(dmg_multi_climate *h, const float *terrain)
{
    dmg_multi *m = reinterpret_cast<dmg_multi *>(h);
    DMG_CHECK_ARG(m != nullptr && !m->cl.empty() && terrain != nullptr);
    return m->team.run([&](int i) { return dmg_climate_set_terrain(m->cl[i], terrain); });
}

// owned rows of this process's bands into the FULL-grid array; synchronises
int dmg_multi_climate_get(dmg_multi_climate *h, float *T)
{
    dmg_multi *m = reinterpret_cast<dmg_multi *>(h);
    DMG_CHECK_ARG(m != nullptr && !m->cl.empty() && T != nullptr);
    return m->team.run([&](int i) -> int { int rc = multi_wait(m, i); return rc ? rc : dmg_climate_get(m->cl[i], T); });
}

int dmg_multi_climate_sync(dmg_multi_climate *h)
{
    dmg_multi *m = reinterpret_cast<dmg_multi *>(h);
    DMG_CHECK_ARG(m != nullptr);
    return m->team.run([&](int i) { return multi_wait(m, i); });
}

void dmg_multi_destroy(dmg_multi *m) { multi_free(m); }

int dmg_multi_band_count(const dmg_multi *m) { return m ? m->n : 0; }

int dmg_multi_band(dmg_multi *m, int local_index, dmg_ocean **out)
{
    DMG_CHECK_ARG(m != nullptr && out != nullptr && local_index >= 0 && local_index < m->n && (int)m->oc.size() == m->n);
    *out = m->oc[local_index];
    return DMG_OK;
}

int dmg_multi_set_order(dmg_multi *m, int order)
{
    DMG_CHECK_ARG(m != nullptr && (order == DMG_MULTI_EXCHANGE_THEN_SWEEP || order == DMG_MULTI_OVERLAP_LAST_LAUNCH));
    m->order = order;
    return DMG_OK;
}

#define MULTI_FOR_BANDS(call)                                             \
    DMG_CHECK_ARG(m != nullptr && (int)m->oc.size() == m->n);             \
    return m->team.run([&](int i) -> int { dmg_ocean *o = m->oc[i]; (void)o; return call; })

int dmg_multi_set_terrain(dmg_multi *m, const float *img) { DMG_CHECK_ARG(img != nullptr); MULTI_FOR_BANDS(dmg_ocean_set_terrain(o, img)); }
int dmg_multi_set_divw(dmg_multi *m, const float *divw) { DMG_CHECK_ARG(divw != nullptr); MULTI_FOR_BANDS(dmg_ocean_set_divw(o, divw)); }
int dmg_multi_set_pressure(dmg_multi *m, const float *p) { DMG_CHECK_ARG(p != nullptr); MULTI_FOR_BANDS(dmg_ocean_set_pressure(o, p)); }
int dmg_multi_set_velocity(dmg_multi *m, const float *vx, const float *vy)
{
    DMG_CHECK_ARG(vx != nullptr && vy != nullptr);
    MULTI_FOR_BANDS(dmg_ocean_set_velocity(o, vx, vy));
}
int dmg_multi_generate_terrain(dmg_multi *m, const dmg_noise_params *params)
{
    DMG_CHECK_ARG(params != nullptr);
    MULTI_FOR_BANDS(dmg_ocean_generate_terrain(o, params));
}
int dmg_multi_get(dmg_multi *m, float *vx, float *vy, float *pressure, float *divw) { MULTI_FOR_BANDS(dmg_ocean_get(o, vx, vy, pressure, divw)); }
int dmg_multi_sync(dmg_multi *m) { DMG_CHECK_ARG(m != nullptr); return m->team.run([&](int i) { return multi_wait(m, i); }); }

int dmg_multi_jacobi(dmg_multi *m, int iters, int zero_first)
{
    DMG_CHECK_ARG(m != nullptr && iters >= 0 && (int)m->oc.size() == m->n);
    if (m->dead.load()) return dmg_fail(DMG_ESTATE, "this dmg_multi has lost a band: destroy it");
    return m->team.run([&](int i) { return band_jacobi(m, i, iters, zero_first); });
}

int dmg_multi_step(dmg_multi *m, int n_outer)
{
    DMG_CHECK_ARG(m != nullptr && n_outer >= 0 && (int)m->oc.size() == m->n);
    if (m->dead.load()) return dmg_fail(DMG_ESTATE, "this dmg_multi has lost a band: destroy it");
    std::vector<int> scratch((size_t)m->n, 0);
    return m->team.run([&](int i) { return band_step(m, i, n_outer, scratch); });
}

int dmg_multi_exchange(dmg_multi *m, int field, int rows)
{
    DMG_CHECK_ARG(m != nullptr && rows >= 1 && rows <= m->ghost && (int)m->oc.size() == m->n);
    if (m->world == 1) return DMG_OK;
    return m->team.run([&](int i) { return band_exchange(m, i, field, rows); });
}

int dmg_multi_get_stats(dmg_multi *m, dmg_multi_stats *s)
{
    DMG_CHECK_ARG(m != nullptr && s != nullptr);
    int rc = dmg_multi_sync(m);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(m->ctx_of(0)->device));
    if ((rc = multi_fold_events(m))) return rc;
    memset(s, 0, sizeof *s);
    s->bands_total = m->world; s->bands_local = m->n; s->transport = m->transport; s->order = m->order; s->ghost = m->ghost;
    s->exchanges = m->n_exchanges; s->exchange_ms = m->exchange_ms; s->exchange_bytes = m->exchange_bytes;
    long long l = 0, sw = 0;
    if (!m->oc.empty()) (void)dmg_ocean_counters(m->oc[0], &l, &sw);
    s->jacobi_launches = l; s->jacobi_sweeps = sw;
    return DMG_OK;
}

int dmg_multi_timer_start(dmg_multi *m) { DMG_CHECK_ARG(m != nullptr); return dmg_timer_start(m->ctx_of(0)); }
int dmg_multi_timer_stop(dmg_multi *m, float *ms)
{
    DMG_CHECK_ARG(m != nullptr && ms != nullptr);
    int rc = dmg_multi_sync(m);                              // every band, not only the one that carries the timer
    if (rc) return rc;
    return dmg_timer_stop(m->ctx_of(0), ms);
}

}  // extern "C"
