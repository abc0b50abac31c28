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
// Transports of the halo slabs (dmg_ocean_halo: contiguous rows):
//   RCCL            grouped ncclSend/ncclRecv on the band's stream: one RCCL kernel per exchange
//   RCCL loopback   the same calls towards the process's own rank (tests on a one-GPU box)
//   peer copies     hipMemcpyPeerAsync pulled by the receiver, ordered by events and two host barriers (one process)
//   store           direct stores ("peer-mapped direct stores for the halo with RCCL kept for setup and
//                   reductions"): a copy kernel of the SENDER writes its edge rows straight into a landing slot in the
//                   neighbour's memory -- a peer-mapped pointer in one process, a hipIpc mapping between processes -- and then
//                   a sequence number into the neighbour's flag word; the receiver's stream runs a kernel that waits for that
//                   number and moves the slot into its ghost rows.  No RCCL kernel, no host barrier, no event between the
//                   GPUs: what orders the two streams is the flag.  Two slots per side, used alternately: a sender can only be
//                   one exchange ahead of its neighbour, because its next exchange needs the neighbour's rows of this one.
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
    // A band that fails with a runtime error calls on_fail from its own thread: it may only MARK the handle (an atomic flag
    // that the other bands' waits poll).  Whatever tears shared state down -- ncclCommAbort, which the siblings may be inside
    // of at that moment -- runs in after_fail, on the caller's thread, once every band thread has returned.
    std::function<void(int)> on_fail;
    std::function<void()> after_fail;
    bool runtime_failure = false;

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
                    const bool runtime = r == DMG_EHIP || r == DMG_ENOMEM;   // (a refused argument is collective and leaves nothing queued)
                    if (runtime && on_fail) on_fail(i);
                    {
                        std::lock_guard<std::mutex> lk(mtx);
                        rc[i] = r;
                        err[i] = e;
                        if (runtime) runtime_failure = true;
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
            if (r == DMG_EHIP || r == DMG_ENOMEM) {
                const std::string keep = dmg_last_error();
                if (on_fail) on_fail(0);
                if (after_fail) after_fail();
                return dmg_fail(r, "%s", keep.c_str());
            }
            return r;
        }
        {
            std::lock_guard<std::mutex> lk(mtx);
            job = f;
            remaining = n;
            failed = false;
            runtime_failure = false;
            bar_count = 0;
            generation++;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(mtx);
        cv_done.wait(lk, [&] { return remaining == 0; });
        if (runtime_failure && after_fail) { lk.unlock(); after_fail(); lk.lock(); }     // every band thread is parked again
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

// ---- the store transport's per-band state ----
// my landing slots [side][slot] and flag words [side] (what the neighbours write), the neighbours' (what I write)
struct store_side {
    unsigned char *land = nullptr;       // my two landing slots for this side, slot_bytes each
    unsigned *flag = nullptr;            // my flag word for this side: the sequence number of the last slab that has landed whole
    unsigned char *r_land = nullptr;     // the neighbour's landing slots for ITS side that faces me
    unsigned *r_flag = nullptr;
    unsigned *counter = nullptr;         // device word: blocks of a push kernel that are done
    unsigned seq = 0;                    // slabs exchanged over this boundary so far (both neighbours count alike)
};
struct store_band {
    unsigned char *land_base = nullptr;  // one allocation: 2 sides x 2 slots
    unsigned *flag_base = nullptr;       // fine-grained (uncached) device memory: 2 flag words, 256 B apart
    unsigned *counters = nullptr;
    int *err_host = nullptr;             // host-mapped: a landing kernel gave up waiting for its flag
    int *err_dev = nullptr;
    void *ipc_land[2] = { nullptr, nullptr }, *ipc_flag[2] = { nullptr, nullptr };   // mappings opened with hipIpcOpenMemHandle (one process per GPU)
    store_side side[2];
};
// what a rank publishes so that its neighbours can map its landing slots and flags (dmg_multi_store_blob / _connect)
struct store_blob {
    hipIpcMemHandle_t land, flag;
    unsigned long long slot_bytes;
    int rank, device;
    char pad[DMG_MULTI_STORE_BLOB_BYTES - 2 * sizeof(hipIpcMemHandle_t) - sizeof(unsigned long long) - 2 * sizeof(int)];
};
static_assert(sizeof(store_blob) == DMG_MULTI_STORE_BLOB_BYTES, "store_blob layout");

// progress of a band's exchanges: an event per exchange in a ring; a host-side wait only gives up when no new one completes
struct band_beat {
    enum { RING = 128 };
    hipEvent_t ring[RING] = {};
    long long issued = 0, seen = 0;
};

struct dmg_multi {
    int world = 1;                   // bands in the whole job
    int first = 0;                   // global index of this process's first band
    int n = 0;                       // bands in this process
    int transport = DMG_MULTI_PEER;
    int order = DMG_MULTI_EXCHANGE_THEN_SWEEP;
    int ghost = 1;
    int warnings = 0;                // DMG_MULTI_WARN_*
    bool shared_ctx = false;         // loopback: every band on ctx[0] (one stream)
    bool same_device = false;        // one process, and two neighbouring bands share a GPU (tests, rehearsals)
    dmg_grid grid{};
    dmg_ocean_params prm{};
    std::vector<dmg_ctx *> ctx;
    std::vector<dmg_ocean *> oc;     // the bands of an ocean (dmg_multi_create*), or ...
    std::vector<dmg_climate *> cl;   // ... the bands of a climate field (dmg_multi_climate_create*): same streams, threads and transports
    std::vector<ncclComm_t> comm;    // RCCL: one per local band (loopback: one in all); store, one process per GPU: set-up and the reach all-reduce
    std::vector<hipEvent_t> ev_done, ev_copied;    // PEER: per band
    // DMG_MULTI_OVERLAP_LAST_LAUNCH: a second, high-priority stream per band carries the exchange while the band's own
    // stream sweeps the interior units of the block's last launch.  Bands of one process that share a GPU share that stream
    // (comm_owner[i] = the band that owns band i's): a device then holds n + 1 streams instead of 2 n
    std::vector<hipStream_t> comm_stream;
    std::vector<int> comm_owner;
    std::vector<hipEvent_t> ev_edge, ev_xchg;      // per band: a whole launch done (own stream) / ghost rows refreshed (second stream)
    std::atomic<bool> dead{false};   // a band failed: only dmg_multi_destroy remains (the communicators are aborted once the band threads have drained)
    int timeout_ms = 120000;         // how long a host-side wait, or a landing kernel, may go without progress before the handle is given up
    int fake_link_us = 0;            // experiments (DMG_MULTI_FAKE_LINK_US=<us>): every exchange is followed, on its stream, by a kernel that
                                     // waits that long -- a link slower than a local copy, to see what the overlapped order hides of it
    int fail_band = -1;              // tests (DMG_MULTI_FAIL=<band>:<n>[:<order>]): that band gives up in front of its n-th exchange
    long long fail_at = 0;
    int fail_order = -1;             //   ... only while the handle runs this exchange order (-1: any)
    std::vector<long long> n_band_exchanges;
    std::vector<int *> d_red;        // per band: one device word for reductions
    std::vector<store_band> store;   // DMG_MULTI_STORE: per local band
    size_t slot_bytes = 0;
    bool store_connected = false;
    std::vector<band_beat> beat;
    band_team team;
    // halo-exchange time: event pairs on band 0's stream around every exchange, folded into the sum when read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free, ev_used;
    double exchange_ms = 0.0;
    long long n_exchanges = 0, exchange_bytes = 0;
    dmg_ctx *ctx_of(int i) const { return shared_ctx ? ctx[0] : ctx[i]; }
    hipStream_t cs_of(int i) const { return comm_stream[shared_ctx ? 0 : comm_owner[i]]; }
    bool uses_rccl_data() const { return transport == DMG_MULTI_RCCL || transport == DMG_MULTI_RCCL_LOOPBACK; }
};

static bool multi_trace() { static const bool on = dmg_hook("DMG_MULTI_TRACE") != nullptr; return on; }
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

// A band has failed, or a wait has run out.  From a band's thread: only the flag -- the other bands' waits poll it and return.
static void multi_mark_dead(dmg_multi *m) { m->dead.store(true); }
// Once every band thread has returned (band_team::after_fail): abort the local communicators, so that receives still queued
// on the streams end instead of waiting for a peer that will never send (ncclCommAbort tears the communicator's kernels
// down).  The bands of other processes notice through their own timed waits (multi_wait).
static void multi_abort_comms(dmg_multi *m)
{
    m->dead.store(true);
    for (ncclComm_t &c : m->comm)
        if (c) { (void)ncclCommAbort(c); c = nullptr; }
}

// Host-side wait for band i: until its exchange number `target` has completed (target > 0), or until its stream is idle
// (target == 0).  With RCCL a peer that has died leaves a receive kernel spinning forever, and the store transport's landing
// kernels give up by themselves and say so in a host-mapped word: the wait polls, watches both, and gives up when NO PROGRESS
// has been made for timeout_ms (DMG_MULTI_TIMEOUT_MS) -- progress being a newly completed exchange of this band (band_beat),
// not wall time: a long queue of asynchronous work (dmg_multi_climate_steps: 500 000 steps) is not a failure.
static int multi_wait_for(dmg_multi *m, int i, long long target)
{
    dmg_ctx *c = m->ctx_of(i);
    DMG_HIP(hipSetDevice(c->device));
    band_beat &b = m->beat[i];
    const bool watched = (m->uses_rccl_data() && !m->comm.empty()) || m->transport == DMG_MULTI_STORE;
    if (!watched && target == 0) { DMG_HIP(hipStreamSynchronize(c->stream)); return DMG_OK; }
    auto t0 = std::chrono::steady_clock::now();
    ncclComm_t comm = m->comm.empty() ? nullptr : m->comm[m->transport == DMG_MULTI_RCCL_LOOPBACK ? 0 : i];
    for (long long spin = 0;; spin++) {
        while (b.seen < b.issued) {                          // exchanges complete in issue order per stream
            const hipError_t q = hipEventQuery(b.ring[(b.seen + 1) % band_beat::RING]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return dmg_fail(DMG_EHIP, "band %d: an exchange failed: %s", m->first + i, hipGetErrorString(q));
            b.seen++;
            t0 = std::chrono::steady_clock::now();
        }
        if (target > 0) { if (b.seen >= target) return DMG_OK; }
        else {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return dmg_fail(DMG_EHIP, "band %d: stream failed: %s", m->first + i, hipGetErrorString(q));
        }
        if (m->dead.load()) return dmg_fail(DMG_ESTATE, "stopped: this dmg_multi has lost a band");
        if ((spin & 63) == 63) {
            if (m->transport == DMG_MULTI_STORE && (int)m->store.size() > i && m->store[i].err_host && *(volatile int *)m->store[i].err_host) {
                multi_mark_dead(m);
                return dmg_fail(DMG_EHIP, "band %d: a neighbour's rows did not arrive within %d ms (store transport; the neighbour band has "
                                          "probably failed)", m->first + i, m->timeout_ms);
            }
            ncclResult_t async = ncclSuccess;
            if (watched && m->uses_rccl_data() && comm && ncclCommGetAsyncError(comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                multi_mark_dead(m);
                return dmg_fail(DMG_EHIP, "band %d: RCCL reports %s; communicators aborted", m->first + i, ncclGetErrorString(async));
            }
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (watched && ms > m->timeout_ms) {
                multi_mark_dead(m);
                return dmg_fail(DMG_EHIP, "band %d: no progress for %d ms (a neighbour band has probably failed); communicators aborted "
                                          "(DMG_MULTI_TIMEOUT_MS)", m->first + i, m->timeout_ms);
            }
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
        else std::this_thread::yield();
    }
    if (m->transport == DMG_MULTI_STORE && (int)m->store.size() > i && m->store[i].err_host && *(volatile int *)m->store[i].err_host) {
        multi_mark_dead(m);
        return dmg_fail(DMG_EHIP, "band %d: a neighbour's rows did not arrive within %d ms (store transport; the neighbour band has probably "
                                  "failed)", m->first + i, m->timeout_ms);
    }
    return DMG_OK;
}
static int multi_wait(dmg_multi *m, int i) { return multi_wait_for(m, i, 0); }

// one more exchange of band i has been queued on `st`: its event goes into the ring.  The ring never holds two records of one
// slot in flight -- a host thread more than RING exchanges ahead of its GPU waits here (with the same watchdog) -- so every
// slot between `seen` and `issued` belongs to exactly one exchange and multi_wait_for can tell progress from a hang
static int band_beat_record(dmg_multi *m, int i, hipStream_t st)
{
    band_beat &b = m->beat[i];
    const long long q = b.issued + 1;
    if (q > band_beat::RING) { int rc = multi_wait_for(m, i, q - band_beat::RING); if (rc) return rc; }
    hipEvent_t &e = b.ring[q % band_beat::RING];
    if (!e) DMG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    DMG_HIP(hipEventRecord(e, st));
    b.issued = q;
    return DMG_OK;
}

// experiments: one wave that holds its stream for `ticks` of the 100 MHz clock (DMG_MULTI_FAKE_LINK_US)
__global__ void k_multi_fake_link(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// ---- store transport: the two kernels ----
// push: `n` bytes from my rows into the neighbour's landing slot, then -- when every block's stores have reached the
// neighbour's memory -- the sequence number into its flag word.  dst and flag are peer-mapped (or IPC-mapped) pointers into
// another GPU's memory: the stores travel over xGMI as posted writes.  One launch carries both sides of a band (blockIdx.y).
struct store_job {                       // one side of an exchange, as a kernel argument
    const unsigned char *src; unsigned char *dst; size_t n;
    unsigned *flag; unsigned seq; unsigned *counter;
    unsigned long long ticks; int *err;
};
struct store_jobs { store_job side[2]; };    // blockIdx.y picks the side: both sides of a band travel in ONE launch

__global__ __launch_bounds__(256) void k_store_push(store_jobs jobs)
{
    const store_job &j = jobs.side[blockIdx.y];
    const unsigned char *__restrict__ src = j.src;
    unsigned char *__restrict__ dst = j.dst;
    const size_t n = j.n;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if ((((size_t)src | (size_t)dst | n) & 15) == 0) {
        const uint4 *s4 = (const uint4 *)src; uint4 *d4 = (uint4 *)dst;
        for (size_t k = tid; k < n / 16; k += nthr) d4[k] = s4[k];
    } else {
        const unsigned *s1 = (const unsigned *)src; unsigned *d1 = (unsigned *)dst;      // (rows are whole floats)
        for (size_t k = tid; k < n / 4; k += nthr) d1[k] = s1[k];
    }
    __threadfence_system();                                  // my stores are visible to the neighbour's GPU before ...
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(j.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (done == gridDim.x) {                             // ... the last block of this side to finish publishes the slab
            __hip_atomic_store(j.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(j.flag, j.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// land: wait until the neighbour's slab number `seq` is in the slot, then move it into the ghost rows.  Every block polls the
// flag by itself (one lane, with s_sleep between reads) and gives up after `ticks` of the 100 MHz clock: it then raises the
// band's error word and copies nothing -- every wave reaches its end whatever the neighbour does.
__global__ __launch_bounds__(256) void k_store_land(store_jobs jobs)
{
    const store_job &j = jobs.side[blockIdx.y];
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int good = 1;
        while ((int)(__hip_atomic_load(j.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - j.seq) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > j.ticks) { good = 0; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        if (!good) __hip_atomic_store(j.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    // Who wrote these bytes: another GPU (or this one, in a rehearsal), with plain stores into THIS device's memory over
    // xGMI, released at system scope before the flag.  Who reads them: every wave of this launch, which may have been
    // resident -- and this XCD's L2 may hold lines of the same slot from two exchanges ago -- since before they arrived.
    // So (1) the slot is fine-grained memory (store_alloc): its lines are never kept across a system-scope acquire; (2) every
    // wave performs that acquire itself behind the flag (it invalidates this CU's L1 and the L2's non-coherent lines:
    // buffer_inv sc0 sc1); (3) the loads themselves are system-scope loads (sc0 sc1), which no cache level answers from a
    // line it filled before the acquire.  Any one of the three is what the memory model asks for; the slab is 2 MB.
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const unsigned char *__restrict__ src = j.src;
    unsigned char *__restrict__ dst = j.dst;
    const size_t n = j.n;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if ((((size_t)src | (size_t)dst | n) & 15) == 0) {
        const unsigned long long *s8 = (const unsigned long long *)src; ulonglong2 *d16 = (ulonglong2 *)dst;
        for (size_t k = tid; k < n / 16; k += nthr) {
            ulonglong2 v;
            v.x = __hip_atomic_load(s8 + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v.y = __hip_atomic_load(s8 + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            d16[k] = v;
        }
    } else {
        const unsigned *s1 = (const unsigned *)src; unsigned *d1 = (unsigned *)dst;      // (rows are whole floats)
        for (size_t k = tid; k < n / 4; k += nthr) d1[k] = __hip_atomic_load(s1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static void store_free(dmg_multi *m)
{
    for (size_t i = 0; i < m->store.size(); i++) {
        store_band &sb = m->store[i];
        if (m->ctx_of((int)i)) (void)hipSetDevice(m->ctx_of((int)i)->device);
        for (int s = 0; s < 2; s++) {
            if (sb.ipc_land[s]) (void)hipIpcCloseMemHandle(sb.ipc_land[s]);
            if (sb.ipc_flag[s]) (void)hipIpcCloseMemHandle(sb.ipc_flag[s]);
        }
        if (sb.land_base) (void)hipFree(sb.land_base);
        if (sb.flag_base) (void)hipFree(sb.flag_base);
        if (sb.counters) (void)hipFree(sb.counters);
        if (sb.err_host) (void)hipHostFree(sb.err_host);
    }
    m->store.clear();
}

// landing slots, flags, counters and the error word of local band i (on its device)
static int store_alloc(dmg_multi *m, int i)
{
    store_band &sb = m->store[i];
    DMG_HIP(hipSetDevice(m->ctx_of(i)->device));
    // The landing slots are written by ANOTHER GPU while a kernel of this one is resident and about to read them: that is
    // what fine-grained device memory is for (coarse-grained memory is only coherent at kernel boundaries); uncached device
    // memory is the fall-back, and there is no coarse-grained one
    static const bool coarse_ab = dmg_hook("DMG_STORE_COARSE") != nullptr;      // A/B measurements only: round 4's coarse-grained slots
    if (coarse_ab) { if (hipMalloc((void **)&sb.land_base, 4 * m->slot_bytes) != hipSuccess) return dmg_fail(DMG_ENOMEM, "store transport: hipMalloc failed"); }
    else if (hipExtMallocWithFlags((void **)&sb.land_base, 4 * m->slot_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        sb.land_base = nullptr;
        if (hipExtMallocWithFlags((void **)&sb.land_base, 4 * m->slot_bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            sb.land_base = nullptr;
            return dmg_fail(DMG_ENOMEM, "store transport: no fine-grained or uncached device memory for the landing slots (%zu bytes)", 4 * m->slot_bytes);
        }
    }
    // the flag words likewise: written by another GPU while a kernel of this one polls them
    if (hipExtMallocWithFlags((void **)&sb.flag_base, 4096, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags((void **)&sb.flag_base, 4096, hipDeviceMallocFinegrained) != hipSuccess)
            return dmg_fail(DMG_ENOMEM, "store transport: no fine-grained device memory for the flag words");
    }
    DMG_HIP(hipMemset(sb.flag_base, 0, 4096));
    DMG_HIP(hipMalloc((void **)&sb.counters, 256));
    DMG_HIP(hipMemset(sb.counters, 0, 256));
    DMG_HIP(hipHostMalloc((void **)&sb.err_host, sizeof(int), hipHostMallocMapped));
    *sb.err_host = 0;
    DMG_HIP(hipHostGetDevicePointer((void **)&sb.err_dev, sb.err_host, 0));
    for (int s = 0; s < 2; s++) {
        sb.side[s].land = sb.land_base + (size_t)s * 2 * m->slot_bytes;
        sb.side[s].flag = sb.flag_base + s * 64;
        sb.side[s].counter = sb.counters + s * 16;
    }
    return DMG_OK;
}

// one process: the neighbours' slots and flags are plain pointers (peer access is enabled for bands on distinct GPUs)
static int store_connect_local(dmg_multi *m)
{
    for (int i = 0; i < m->n; i++)
        for (int s = 0; s < 2; s++) {
            const int j = s == DMG_SIDE_SOUTH ? i - 1 : i + 1;
            if (j < 0 || j >= m->n) continue;
            m->store[i].side[s].r_land = m->store[j].side[s ^ 1].land;
            m->store[i].side[s].r_flag = m->store[j].side[s ^ 1].flag;
        }
    m->store_connected = true;
    return DMG_OK;
}

static int store_make_blob(dmg_multi *m, store_blob *b)
{
    memset(b, 0, sizeof *b);
    DMG_HIP(hipSetDevice(m->ctx[0]->device));
    DMG_HIP(hipIpcGetMemHandle(&b->land, m->store[0].land_base));
    DMG_HIP(hipIpcGetMemHandle(&b->flag, m->store[0].flag_base));
    b->slot_bytes = m->slot_bytes; b->rank = m->first; b->device = m->ctx[0]->device;
    return DMG_OK;
}

// one process per GPU: map the two neighbours' landing slots and flags from the blobs of all ranks
static int store_connect_rank(dmg_multi *m, const store_blob *all)
{
    DMG_HIP(hipSetDevice(m->ctx[0]->device));
    store_band &sb = m->store[0];
    for (int s = 0; s < 2; s++) {
        const int j = s == DMG_SIDE_SOUTH ? m->first - 1 : m->first + 1;
        if (j < 0 || j >= m->world) continue;
        const store_blob &nb = all[j];
        if (nb.rank != j || nb.slot_bytes != m->slot_bytes)
            return dmg_fail(DMG_EINVAL, "store transport: rank %d's blob does not match (rank %d, slots of %llu bytes, mine %zu)", j, nb.rank,
                            nb.slot_bytes, m->slot_bytes);
        DMG_HIP(hipIpcOpenMemHandle(&sb.ipc_land[s], nb.land, hipIpcMemLazyEnablePeerAccess));
        DMG_HIP(hipIpcOpenMemHandle(&sb.ipc_flag[s], nb.flag, hipIpcMemLazyEnablePeerAccess));
        sb.side[s].r_land = (unsigned char *)sb.ipc_land[s] + (size_t)(s ^ 1) * 2 * m->slot_bytes;
        sb.side[s].r_flag = (unsigned *)sb.ipc_flag[s] + (s ^ 1) * 64;
    }
    m->store_connected = true;
    return DMG_OK;
}

// ---- one halo exchange of `field` (rows deep) as seen by local band i; every band thread calls it ----
// `on`: the stream that carries it (default: the band's own)
static int band_exchange(dmg_multi *m, int i, int field, int rows, int buffer = -1, hipStream_t on = nullptr)
{
    dmg_ctx *c = m->ctx_of(i);
    const int g = m->first + i;
    MTRACE("band %d: exchange field %d rows %d transport %d", g, field, rows, m->transport);
    if (m->dead.load()) return dmg_fail(DMG_ESTATE, "this dmg_multi has lost a band: destroy it");
    if (g == m->fail_band && (m->fail_order < 0 || m->fail_order == m->order) && ++m->n_band_exchanges[i] == m->fail_at)
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
    } else if (m->transport == DMG_MULTI_STORE) {
        // direct stores: my edge rows into the neighbours' landing slots (push), theirs from my slots into my ghost rows (land).
        // A slab larger than a slot (the advect halo of a fast ocean) travels in pieces; a piece is pushed only after the piece
        // before it has landed here, which is what keeps a sender from overwriting a slot its neighbour has not emptied.
        if (!m->store_connected) return dmg_fail(DMG_ESTATE, "store transport: dmg_multi_store_connect has not been called");
        store_band &sb = m->store[i];
        void *send[2] = { nullptr, nullptr }, *recv[2] = { nullptr, nullptr }; size_t bytes[2] = { 0, 0 };
        for (int side = 0; side < 2; side++) {
            const int peer = side == DMG_SIDE_SOUTH ? g - 1 : g + 1;
            if (peer < 0 || peer >= m->world) continue;
            if ((rc = slab(i, side, &send[side], &recv[side], &bytes[side]))) return rc;
        }
        // (every boundary of the job carries the same number of bytes, so every band walks the same number of pieces -- the
        // barriers below need that; a band at the end of the row has one side only)
        const unsigned long long ticks = (unsigned long long)m->timeout_ms * 100000ull;
        const size_t total = (size_t)rows * m->grid.W * (m->cl.empty() && (field == DMG_F_V || field == DMG_F_VADVECT) ? 8 : 4);
        for (int side = 0; side < 2; side++)
            if (bytes[side] && bytes[side] != total) return dmg_fail(DMG_EINVAL, "store transport: a slab of %zu bytes where %zu were expected", bytes[side], total);
        for (size_t off = 0; off < total; off += m->slot_bytes) {
            const size_t piece = std::min(m->slot_bytes, total - off);
            store_jobs pj{}, lj{};
            int ns = 0;
            for (int side = 0; side < 2; side++) {
                if (!bytes[side]) continue;
                store_side &ss = sb.side[side];
                const unsigned q = ++ss.seq;
                pj.side[ns] = store_job{ (const unsigned char *)send[side] + off, ss.r_land + (size_t)(q & 1) * m->slot_bytes, piece, ss.r_flag, q, ss.counter, 0ull, nullptr };
                lj.side[ns] = store_job{ ss.land + (size_t)(q & 1) * m->slot_bytes, (unsigned char *)recv[side] + off, piece, ss.flag, q, nullptr, ticks, sb.err_dev };
                ns++;
                if (i == 0) m->exchange_bytes += 2 * (long long)piece;
            }
            const unsigned pblocks = (unsigned)std::min<size_t>(128, (piece + 65535) / 65536 * 4), lblocks = (unsigned)std::min<size_t>(64, (piece + 65535) / 65536 * 2);
            if (ns) {
                hipLaunchKernelGGL(k_store_push, dim3(pblocks, (unsigned)ns), dim3(256), 0, st, pj);
                DMG_HIP(hipGetLastError());
            }
            // bands of one process that share a GPU: nobody's landing kernel is queued before everybody's push is -- a landing
            // kernel polls for a push that, on a shared hardware queue, could otherwise sit behind it
            if (m->same_device) TEAM_BARRIER(m);
            if (ns) {
                hipLaunchKernelGGL(k_store_land, dim3(lblocks, (unsigned)ns), dim3(256), 0, st, lj);
                DMG_HIP(hipGetLastError());
            }
            if (m->same_device) TEAM_BARRIER(m);
        }
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
    if (m->transport != DMG_MULTI_PEER && m->world > 1 && (rc = band_beat_record(m, i, st))) return rc;
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
    hipStream_t cs = m->cs_of(i);
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
    if (m->n < m->world && m->comm.empty())
        return dmg_fail(DMG_ESTATE, "this handle was created without an RCCL id: the advect reach cannot be all-reduced (dmg_multi_step needs one)");
    if (m->transport == DMG_MULTI_RCCL || m->n < m->world) {
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
        if (reach > g) {
            // currents have developed (|v| = 10 at H = 8192 reaches ~96 rows, :294-299): this advect gathers from a copy of v
            // that is `reach` rows deep (dmg_ocean_advect_prepare; grown on demand), whose ghost rows travel like any other
            // field's.  Still all or nobody: the reach is the same on every band, and so is the answer to "does it fit"
            if (reach > m->grid.H / m->world)
                return dmg_fail(DMG_EINVAL, "advect reaches %d rows but a band owns only %d: a halo must come from ONE neighbour; use "
                                            "fewer bands (every band stopped before the step, nothing was changed)", reach, m->grid.H / m->world);
            if ((rc = dmg_ocean_advect_prepare(o, reach))) return rc;
            if ((rc = band_exchange(m, i, DMG_F_VADVECT, reach))) return rc;
        } else if ((rc = band_exchange(m, i, DMG_F_V, g))) return rc;
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
    for (size_t i = 0; i < m->comm_stream.size(); i++)       // (nothing of this handle may still be running when its memory goes)
        if (m->comm_stream[i] && m->ctx_of((int)i)) { (void)hipSetDevice(m->ctx_of((int)i)->device); (void)hipStreamSynchronize(m->comm_stream[i]); }
    for (size_t i = 0; i < m->ctx.size(); i++)
        if (m->ctx[i]) { (void)hipSetDevice(m->ctx[i]->device); (void)hipStreamSynchronize(m->ctx[i]->stream); }
    for (dmg_ocean *o : m->oc) if (o) dmg_ocean_destroy(o);
    for (dmg_climate *k : m->cl) if (k) dmg_climate_destroy(k);
    store_free(m);
    for (band_beat &b : m->beat) for (hipEvent_t e : b.ring) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_copied) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_edge) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_xchg) if (e) (void)hipEventDestroy(e);
    for (size_t i = 0; i < m->comm_stream.size(); i++)
        if (m->comm_stream[i] && (m->comm_owner.empty() || m->comm_owner[i] == (int)i)) {
            (void)hipSetDevice(m->ctx_of((int)i)->device); (void)hipStreamSynchronize(m->comm_stream[i]); (void)hipStreamDestroy(m->comm_stream[i]);
        }
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
    if (const char *e = dmg_hook("DMG_MULTI_FAIL")) {          // tests: "<band>:<n>"
        int b = -1, ord = -1; long long k = 0;
        const int got = sscanf(e, "%d:%lld:%d", &b, &k, &ord);
        if (got >= 2 && b >= 0 && k > 0) { m->fail_band = b; m->fail_at = k; m->fail_order = got == 3 ? ord : -1; }
    }
    if (m->order == DMG_MULTI_OVERLAP_LAST_LAUNCH) {
        // The overlapped order runs two streams per band that wait on each other's events once per block.  HIP maps a process's
        // streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read when the runtime initialises; with four the pattern is
        // pathological on this runtime (18 ms instead of 11 for a 2 x 1024-row rehearsal, profiles/r03_overlap_hwqueues.log).
        // The library does not touch its host's environment: the host sets the variable before its first HIP call -- or calls
        // dmg_runtime_init() there -- and a handle that finds fewer than eight queues asked for falls back to exchange-then-sweep
        // and says so (dmg_multi_get_stats: warnings & DMG_MULTI_WARN_HWQ_FALLBACK; stderr once).
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (!q || atoi(q) < 8) {
            m->order = DMG_MULTI_EXCHANGE_THEN_SWEEP;
            m->warnings |= DMG_MULTI_WARN_HWQ_FALLBACK;
            static std::atomic<bool> said{false};
            bool was = false;
            if (said.compare_exchange_strong(was, true))
                fprintf(stderr, "demiurge_hip: DMG_MULTI_OVERLAP_LAST_LAUNCH needs GPU_MAX_HW_QUEUES >= 8 in the environment before HIP initialises "
                                "(dmg_runtime_init); it is %s: using exchange-then-sweep\n", q ? q : "unset");
        }
    }
    if (const char *e = dmg_hook("DMG_MULTI_FAKE_LINK_US")) {
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
    m->comm_owner.assign((size_t)m->n, 0);
    m->n_band_exchanges.assign((size_t)m->n, 0);
    m->beat.resize((size_t)m->n);
    static const bool share = [] { const char *e = dmg_hook("DMG_MULTI_SHARE_COMM_STREAM"); return !e || atoi(e) != 0; }();   // experiments: 0 = one per band
    for (int i = 0; i < m->n; i++) {
        DMG_HIP(hipSetDevice(m->ctx_of(i)->device));
        int owner = i;
        if (share && !m->shared_ctx)
            for (int j = 0; j < i; j++) if (m->ctx[j]->device == m->ctx[i]->device) { owner = j; break; }
        m->comm_owner[i] = m->shared_ctx ? 0 : owner;
        if (i < ns && owner != i) m->comm_stream[i] = m->comm_stream[owner];
        else if (i < ns) {
            int lo = 0, hi = 0;
            DMG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (hi = greatest priority, numerically lowest)
            static const bool normal = [] { const char *e = dmg_hook("DMG_MULTI_COMM_PRIORITY"); return e && atoi(e) == 0; }();   // experiments
            DMG_HIP(hipStreamCreateWithPriority(&m->comm_stream[i], hipStreamNonBlocking, normal ? lo : hi));
        }
        // experiments (DMG_MULTI_EVENT_FLAGS=2): the join event with an agent-scope release too.  The default keeps the
        // system-scope release: what it orders was written by a neighbour GPU or a copy engine
        static const int evf = [] { const char *e = dmg_hook("DMG_MULTI_EVENT_FLAGS"); return e ? atoi(e) : 1; }();
        DMG_HIP(hipEventCreateWithFlags(&m->ev_edge[i], hipEventDisableTiming));
        DMG_HIP(hipEventCreateWithFlags(&m->ev_xchg[i], hipEventDisableTiming | (evf >= 2 ? hipEventReleaseToDevice : 0)));
    }
    m->team.on_fail = [m](int) { multi_mark_dead(m); };
    m->team.after_fail = [m] { multi_abort_comms(m); };
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
    DMG_CHECK_ARG(transport == DMG_MULTI_RCCL || transport == DMG_MULTI_PEER || transport == DMG_MULTI_RCCL_LOOPBACK || transport == DMG_MULTI_STORE);
    std::unique_ptr<dmg_multi, void (*)(dmg_multi *)> m(new (std::nothrow) dmg_multi, multi_free);
    if (!m) return dmg_fail(DMG_ENOMEM, "out of host memory");
    int rc = multi_common(m.get(), grid, params, ghost, flags);
    if (rc) return rc;
    m->world = n_bands; m->first = 0; m->n = n_bands; m->transport = transport;
    m->shared_ctx = transport == DMG_MULTI_RCCL_LOOPBACK;
    for (int i = 0; i + 1 < n_bands; i++) m->same_device = m->same_device || device_ids[i] == device_ids[i + 1];
    {
        // More than four bands of one process on ONE GPU in the overlapped order (a rehearsal of eight devices on one): the
        // device then runs 9 .. 16 streams that wait on each other once per block, and the time of a solve grows with the number
        // of hardware queues they are spread over -- 8 bands x 160 sweeps: 11 ms with GPU_MAX_HW_QUEUES=4, 47 ms with 8, 57 ms
        // with 16, 9 ms in the plain order (profiles/r04_peer_stall.log): the queues outnumber what the GPU's scheduler keeps
        // resident and it time-slices them.  A real device holds two streams.  Such a handle runs exchange-then-sweep and says so.
        int most = 0;
        for (int i = 0; i < n_bands; i++) { int c = 0; for (int j = 0; j < n_bands; j++) c += device_ids[j] == device_ids[i]; most = std::max(most, c); }
        if (most > 4 && m->order == DMG_MULTI_OVERLAP_LAST_LAUNCH && transport != DMG_MULTI_RCCL_LOOPBACK && !dmg_hook("DMG_MULTI_FORCE_OVERLAP")) {
            m->order = DMG_MULTI_EXCHANGE_THEN_SWEEP;
            m->warnings |= DMG_MULTI_WARN_SHARED_GPU_FALLBACK;
        }
    }
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
    if (transport == DMG_MULTI_PEER || transport == DMG_MULTI_STORE)     // let copy engines / kernels reach the neighbour's memory directly
        for (int i = 0; i + 1 < n_bands; i++) {
            const int a = device_ids[i], b = device_ids[i + 1];
            if (a == b) continue;
            int ok = 0;
            (void)hipDeviceCanAccessPeer(&ok, a, b);
            if (ok) {
                (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0);
                (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0);
                (void)hipGetLastError();                     // "already enabled" is fine
            } else if (transport == DMG_MULTI_STORE)
                return dmg_fail(DMG_EHIP, "store transport: device %d cannot access device %d's memory", a, b);
        }
    if (transport == DMG_MULTI_STORE && n_bands > 1) {
        m->store.resize((size_t)n_bands);
        m->slot_bytes = ((size_t)ghost * grid->W * 8 + 255) & ~(size_t)255;     // (v is the widest field: 8 bytes per cell)
        for (int i = 0; i < n_bands; i++) if ((rc = store_alloc(m.get(), i))) return rc;
        if ((rc = store_connect_local(m.get()))) return rc;
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
    *out = nullptr;
    std::unique_ptr<dmg_multi, void (*)(dmg_multi *)> m(new (std::nothrow) dmg_multi, multi_free);
    if (!m) return dmg_fail(DMG_ENOMEM, "out of host memory");
    int rc = multi_common(m.get(), grid, params, ghost, flags);
    if (rc) return rc;
    int transport = flags & DMG_MULTI_TRANSPORT_MASK;
    if (transport == DMG_MULTI_AUTO) transport = DMG_MULTI_RCCL;
    if (transport != DMG_MULTI_RCCL && transport != DMG_MULTI_STORE)
        return dmg_fail(DMG_EINVAL, "one process per GPU: DMG_MULTI_RCCL or DMG_MULTI_STORE (the peer-copy and loopback transports are one-process forms)");
    if (n_ranks > 1 && transport == DMG_MULTI_RCCL && id128 == nullptr) return dmg_fail(DMG_EINVAL, "DMG_MULTI_RCCL needs the id of dmg_multi_unique_id");
    m->world = n_ranks; m->first = rank; m->n = 1; m->transport = transport;
    m->ctx.assign(1, nullptr); m->d_red.assign(1, nullptr);
    m->ev_done.assign(1, nullptr); m->ev_copied.assign(1, nullptr);
    if ((rc = dmg_create(device_id, nullptr, DMG_PRIVATE_STREAM, &m->ctx[0]))) return rc;
    if ((rc = make_band(m.get(), 0, rank, n_ranks))) return rc;
    DMG_HIP(hipSetDevice(device_id));
    DMG_HIP(hipMalloc((void **)&m->d_red[0], sizeof(int)));
    if (n_ranks > 1 && id128 != nullptr) {
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        m->comm.assign(1, nullptr);
        DMG_NCCL(ncclCommInitRank(&m->comm[0], n_ranks, id, rank));
    }
    if (transport == DMG_MULTI_STORE && n_ranks > 1) {
        // landing slots and flags of this rank; the neighbours map them through hipIpc handles.  With an RCCL communicator the
        // handles travel by one ncclAllGather here (RCCL stays for set-up and for the reach all-reduce); without one (id128 ==
        // NULL: a host with its own bootstrap, or two processes on one GPU in the tests) the host all-gathers
        // dmg_multi_store_blob's bytes itself and hands them to dmg_multi_store_connect
        m->store.resize(1);
        m->slot_bytes = ((size_t)ghost * grid->W * 8 + 255) & ~(size_t)255;     // (v is the widest field: 8 bytes per cell)
        if ((rc = store_alloc(m.get(), 0))) return rc;
        if (!m->comm.empty()) {
            store_blob mine;
            if ((rc = store_make_blob(m.get(), &mine))) return rc;
            unsigned char *d_all = nullptr;
            DMG_HIP(hipMalloc((void **)&d_all, sizeof(store_blob) * (size_t)n_ranks));
            std::vector<store_blob> all((size_t)n_ranks);
            hipStream_t st = m->ctx[0]->stream;
            hipError_t e = hipMemcpyAsync(d_all + sizeof(store_blob) * (size_t)rank, &mine, sizeof mine, hipMemcpyHostToDevice, st);
            ncclResult_t r = e == hipSuccess ? ncclAllGather(d_all + sizeof(store_blob) * (size_t)rank, d_all, sizeof(store_blob), ncclChar, m->comm[0], st) : ncclSuccess;
            if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(all.data(), d_all, sizeof(store_blob) * (size_t)n_ranks, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(st);
            (void)hipFree(d_all);
            if (r != ncclSuccess) return dmg_fail(DMG_EHIP, "store transport: ncclAllGather of the handles failed: %s", ncclGetErrorString(r));
            if (e != hipSuccess) return dmg_fail(DMG_EHIP, "store transport: handle exchange failed: %s", hipGetErrorString(e));
            if ((rc = store_connect_rank(m.get(), all.data()))) return rc;
        }
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
int dmg_multi_climate_set_params(dmg_multi_climate *h, const dmg_climate_params *params)
{
    dmg_multi *m = reinterpret_cast<dmg_multi *>(h);
    DMG_CHECK_ARG(m != nullptr && params != nullptr && !m->cl.empty());
    return m->team.run([&](int i) { return dmg_climate_set_params(m->cl[i], params); });
}

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

// ---- BASELINE config 5 on N GPUs -----------------------------------------------------------------
// dmg_world's sequence -- Tectonics::run steps -> cpufilter::run rounds -> Temperature::run steps (world.hip) -- with the
// stage that can use a node spread over it: the climate year cycle (Temperature.cpp:140-150: 500 000 steps) runs in latitude
// bands through dmg_multi_climate's threads, streams and transports.  Tectonics' rotated gathers and FlowFilter's lake graph
// are global ("replicas only"):
//   one process (dmg_multi_world_create)        they run once, on the first band's GPU; every band pulls ITS rows of the eroded
//                                               terrain from there with one peer copy per iteration (the climate reads the
//                                               terrain's sign only, Temperature.cpp:131-132) behind an event
//   one process per GPU (.._create_rank)        every rank runs them on its own GPU -- replicas, bit-identical by determinism,
//                                               nothing to broadcast -- and owns one climate band
struct dmg_multi_world {
    dmg_ctx *ctx0 = nullptr;
    dmg_grid grid{};
    dmg_tectonics *tect = nullptr;
    float *d_terrain = nullptr, *d_eroded = nullptr;
    dmg_multi *clim = nullptr;
    hipEvent_t ev_ready = nullptr;                   // the eroded terrain of this iteration is complete (ctx0's stream)
    std::vector<hipEvent_t> ev_taken;                // band i has copied its rows (its stream): the next iteration may overwrite them
    long long climate_step = 0, iterations = 0;
    dmg_climate_params clim_prm{};                   // what the climate bands hold
    bool clim_prm_set = false;
};

static void multi_world_free(dmg_multi_world *w)
{
    if (!w) return;
    if (w->clim) multi_free(w->clim);
    if (w->tect) dmg_tectonics_destroy(w->tect);
    if (w->ctx0) {
        (void)hipSetDevice(w->ctx0->device);
        (void)hipStreamSynchronize(w->ctx0->stream);
        if (w->d_terrain) (void)hipFree(w->d_terrain);
        if (w->d_eroded) (void)hipFree(w->d_eroded);
        if (w->ev_ready) (void)hipEventDestroy(w->ev_ready);
        for (hipEvent_t e : w->ev_taken) if (e) (void)hipEventDestroy(e);
        dmg_destroy(w->ctx0);
    }
    delete w;
}

static int multi_world_common(dmg_multi_world *w, int device, const dmg_grid *grid, const float *terrain)
{
    int rc = dmg_create(device, nullptr, DMG_PRIVATE_STREAM, &w->ctx0);
    if (rc) return rc;
    w->grid = *grid;
    const size_t n = (size_t)grid->W * grid->H;
    DMG_HIP(hipSetDevice(device));
    if (hipMalloc((void **)&w->d_terrain, n * 4) != hipSuccess || hipMalloc((void **)&w->d_eroded, n * 4) != hipSuccess)
        return dmg_fail(DMG_ENOMEM, "hipMalloc of two %zu-cell fields failed", n);
    DMG_HIP(hipMemcpyAsync(w->d_terrain, terrain, n * 4, hipMemcpyHostToDevice, w->ctx0->stream));
    DMG_HIP(hipMemcpyAsync(w->d_eroded, w->d_terrain, n * 4, hipMemcpyDeviceToDevice, w->ctx0->stream));
    DMG_HIP(hipStreamSynchronize(w->ctx0->stream));
    DMG_HIP(hipEventCreateWithFlags(&w->ev_ready, hipEventDisableTiming));
    return dmg_tectonics_create(w->ctx0, grid, terrain, &w->tect);
}

int dmg_multi_world_create(int n_bands, const int *device_ids, const dmg_grid *grid, const float *terrain, const float *T0, int ghost, int flags,
                           dmg_multi_world **out)
{
    DMG_CHECK_ARG(out != nullptr && n_bands >= 1 && device_ids != nullptr && grid != nullptr && terrain != nullptr && T0 != nullptr);
    *out = nullptr;
    std::unique_ptr<dmg_multi_world, void (*)(dmg_multi_world *)> w(new (std::nothrow) dmg_multi_world, multi_world_free);
    if (!w) return dmg_fail(DMG_ENOMEM, "out of host memory");
    int rc = multi_world_common(w.get(), device_ids[0], grid, terrain);
    if (rc) return rc;
    if ((rc = multi_create_local(n_bands, device_ids, grid, nullptr, ghost, flags, climate_band(grid, T0, terrain, ghost), &w->clim))) return rc;
    for (int i = 0; i < n_bands; i++) {                        // the bands read the first band's GPU
        if (device_ids[i] == device_ids[0]) continue;
        int ok = 0;
        (void)hipDeviceCanAccessPeer(&ok, device_ids[i], device_ids[0]);
        if (ok) { (void)hipSetDevice(device_ids[i]); (void)hipDeviceEnablePeerAccess(device_ids[0], 0); (void)hipGetLastError(); }
    }
    w->ev_taken.assign((size_t)n_bands, nullptr);
    for (int i = 0; i < n_bands; i++) {
        DMG_HIP(hipSetDevice(device_ids[i]));
        DMG_HIP(hipEventCreateWithFlags(&w->ev_taken[i], hipEventDisableTiming));
    }
    *out = w.release();
    return DMG_OK;
}

int dmg_multi_world_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const float *terrain, const float *T0,
                                int ghost, int flags, dmg_multi_world **out)
{
    DMG_CHECK_ARG(out != nullptr && grid != nullptr && terrain != nullptr && T0 != nullptr);
    *out = nullptr;
    std::unique_ptr<dmg_multi_world, void (*)(dmg_multi_world *)> w(new (std::nothrow) dmg_multi_world, multi_world_free);
    if (!w) return dmg_fail(DMG_ENOMEM, "out of host memory");
    int rc = multi_world_common(w.get(), device_id, grid, terrain);
    if (rc) return rc;
    if ((rc = multi_create_rank(rank, n_ranks, id128, device_id, grid, nullptr, ghost, flags, climate_band(grid, T0, terrain, ghost), &w->clim))) return rc;
    w->ev_taken.assign(1, nullptr);
    DMG_HIP(hipSetDevice(device_id));
    DMG_HIP(hipEventCreateWithFlags(&w->ev_taken[0], hipEventDisableTiming));
    *out = w.release();
    return DMG_OK;
}

void dmg_multi_world_destroy(dmg_multi_world *w) { multi_world_free(w); }

int dmg_multi_world_step(dmg_multi_world *w, const dmg_world_params *params)
{
    DMG_CHECK_ARG(w != nullptr);
    dmg_world_params p;
    if (params) p = *params; else dmg_world_params_default(&p);
    DMG_CHECK_ARG(p.tectonics_steps >= 0 && p.erosion_rounds >= 0 && p.climate_steps >= 0 && p.erosion_N >= 1);
    dmg_multi *m = w->clim;
    if (m->dead.load()) return dmg_fail(DMG_ESTATE, "this handle has lost a band: destroy it");
    DMG_HIP(hipSetDevice(w->ctx0->device));
    hipStream_t st = w->ctx0->stream;
    const size_t bytes = (size_t)w->grid.W * w->grid.H * 4;
    int rc;
    // the stages that do not shard (world.hip has the same sequence; tectonics/Tectonics.cpp:156-275, cpufilter.cpp:41-222)
    if (p.tectonics_steps > 0) {
        if ((rc = dmg_tectonics_step(w->tect, p.tectonics_steps))) return rc;
        DMG_HIP(hipMemcpyAsync(w->d_terrain, tectonics_terrain_device(w->tect), bytes, hipMemcpyDeviceToDevice, st));
    }
    for (hipEvent_t e : w->ev_taken)                     // every band has taken its rows of the iteration before
        if (w->iterations > 0 && e) DMG_HIP(hipStreamWaitEvent(st, e, 0));
    if (p.erosion_rounds > 0) {
        dmg_erosion *e = nullptr;
        if ((rc = erosion_create_from(w->ctx0, &w->grid, w->d_terrain, true, p.erosion_N, p.exponent, p.slope_exponent, p.factor, p.lakes, &e))) return rc;
        rc = dmg_erosion_rounds(e, p.erosion_rounds);
        if (!rc && hipMemcpyAsync(w->d_eroded, erosion_heights_device(e), bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
            rc = dmg_fail(DMG_EHIP, "device copy of the eroded heights failed");
        dmg_erosion_destroy(e);
        if (rc) return rc;
    } else DMG_HIP(hipMemcpyAsync(w->d_eroded, w->d_terrain, bytes, hipMemcpyDeviceToDevice, st));
    DMG_HIP(hipEventRecord(w->ev_ready, st));
    // the climate's bands: each pulls its rows of the eroded terrain (behind the event), then steps; asynchronous
    if (p.climate_steps > 0) {
        if (!w->clim_prm_set) { dmg_climate_params_default(&w->clim_prm); w->clim_prm_set = true; }
        if (memcmp(&p.climate, &w->clim_prm, sizeof p.climate) != 0) {      // other literals than the bands hold: hand them on, once
            if ((rc = dmg_multi_climate_set_params(reinterpret_cast<dmg_multi_climate *>(m), &p.climate))) return rc;
            w->clim_prm = p.climate;
        }
        const int first = (int)w->climate_step, n_steps = p.climate_steps;
        rc = m->team.run([&](int i) -> int {
            dmg_climate *k = m->cl[i];
            int r = climate_set_terrain_peer(k, w->d_eroded, w->ctx0->device, w->ev_ready);
            if (r) return r;
            DMG_HIP(hipEventRecord(w->ev_taken[i], m->ctx_of(i)->stream));
            if (m->world == 1) return dmg_climate_steps(k, first, n_steps);
            for (int done = 0; done < n_steps;) {
                const int n = std::min(m->ghost, n_steps - done);
                if ((r = band_exchange(m, i, 0, m->ghost))) return r;
                if ((r = dmg_climate_steps(k, first + done, n))) return r;
                done += n;
            }
            return DMG_OK;
        });
        if (rc) return rc;
        w->climate_step += p.climate_steps;
    }
    w->iterations++;
    return DMG_OK;
}

// terrain / eroded: full-grid arrays from the GPU that computed them; T: the owned rows of this process's bands
int dmg_multi_world_get(dmg_multi_world *w, float *terrain, float *eroded, float *T)
{
    DMG_CHECK_ARG(w != nullptr);
    DMG_HIP(hipSetDevice(w->ctx0->device));
    const size_t bytes = (size_t)w->grid.W * w->grid.H * 4;
    if (terrain) DMG_HIP(hipMemcpyAsync(terrain, w->d_terrain, bytes, hipMemcpyDeviceToHost, w->ctx0->stream));
    if (eroded) DMG_HIP(hipMemcpyAsync(eroded, w->d_eroded, bytes, hipMemcpyDeviceToHost, w->ctx0->stream));
    DMG_HIP(hipStreamSynchronize(w->ctx0->stream));
    if (T) return dmg_multi_climate_get(reinterpret_cast<dmg_multi_climate *>(w->clim), T);
    return DMG_OK;
}

// store transport, one process per GPU, no RCCL communicator (id128 == NULL at creation): the host carries the handles.
// Every rank: _blob -> all-gather the DMG_MULTI_STORE_BLOB_BYTES bytes of all ranks, in rank order -> _connect.
int dmg_multi_store_blob(dmg_multi *m, void *blob)
{
    DMG_CHECK_ARG(m != nullptr && blob != nullptr);
    if (m->transport != DMG_MULTI_STORE || m->n != 1 || m->store.empty()) return dmg_fail(DMG_ESTATE, "dmg_multi_store_blob: a DMG_MULTI_STORE rank of a job of several");
    store_blob b;
    int rc = store_make_blob(m, &b);
    if (rc) return rc;
    memcpy(blob, &b, sizeof b);
    return DMG_OK;
}

int dmg_multi_store_connect(dmg_multi *m, const void *blobs, int n_ranks)
{
    DMG_CHECK_ARG(m != nullptr && blobs != nullptr && n_ranks == m->world);
    if (m->transport != DMG_MULTI_STORE || m->n != 1 || m->store.empty()) return dmg_fail(DMG_ESTATE, "dmg_multi_store_connect: a DMG_MULTI_STORE rank of a job of several");
    if (m->store_connected) return dmg_fail(DMG_ESTATE, "dmg_multi_store_connect: already connected");
    std::vector<store_blob> all((size_t)n_ranks);
    memcpy(all.data(), blobs, sizeof(store_blob) * (size_t)n_ranks);
    return store_connect_rank(m, all.data());
}

// the climate handles share everything with the ocean's: the same two calls under their names
int dmg_multi_climate_store_blob(dmg_multi_climate *h, void *blob) { return dmg_multi_store_blob(reinterpret_cast<dmg_multi *>(h), blob); }
int dmg_multi_climate_store_connect(dmg_multi_climate *h, const void *blobs, int n_ranks) { return dmg_multi_store_connect(reinterpret_cast<dmg_multi *>(h), blobs, n_ranks); }

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
int dmg_multi_get(dmg_multi *m, float *vx, float *vy, float *pressure, float *divw)
{
    // (the watched wait first: dmg_ocean_get synchronises its stream without a deadline)
    DMG_CHECK_ARG(m != nullptr && (int)m->oc.size() == m->n);
    return m->team.run([&](int i) -> int { int rc = multi_wait(m, i); return rc ? rc : dmg_ocean_get(m->oc[i], vx, vy, pressure, divw); });
}
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
    s->warnings = m->warnings;
    s->exchanges = m->n_exchanges; s->exchange_ms = m->exchange_ms; s->exchange_bytes = m->exchange_bytes;
    long long l = 0, sw = 0;
    if (!m->oc.empty()) (void)dmg_ocean_counters(m->oc[0], &l, &sw);
    s->jacobi_launches = l; s->jacobi_sweeps = sw;
    return DMG_OK;
}

int dmg_multi_comm_ranks(dmg_multi *m, int *ranks, int *devices)
{
    DMG_CHECK_ARG(m != nullptr && ranks != nullptr);
    // what RCCL itself says of the communicator the slabs travel on (ncclCommCount), not what the handle was asked for
    int n = m->world, distinct = 0;
    if (!m->comm.empty() && m->comm[0] && m->transport != DMG_MULTI_RCCL_LOOPBACK) DMG_NCCL(ncclCommCount(m->comm[0], &n));
    for (int i = 0; i < m->n; i++) {
        bool seen = false;
        for (int k = 0; k < i; k++) seen = seen || m->ctx_of(k)->device == m->ctx_of(i)->device;
        distinct += seen ? 0 : 1;
    }
    *ranks = n;
    if (devices) *devices = distinct;
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
