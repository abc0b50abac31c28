// The streaming Jacobi kernels (device code shared with the deep form: ocean_stream_dev.h) and their launcher; the cut of a launch
// into units is host code of its own (ocean_jacobi_stream_plan.hip).
#include "ocean_stream_dev.h"
#include "ocean_stream_plan.h"
#include <algorithm>
#include <chrono>


// one wave per unit; two waves per SIMD leave 256 registers each: all issue slots of a SIMD belong to two fat waves
// Units that stream SOUTHWARDS (round 4).  The kernel is bound by its bytes (above), and a unit of n rows reads n + 16: the
// eight rows beyond each end are read again by the unit above and the unit below -- 9 % of a 180-row unit's reads, 36 % of a
// 28-row unit's (a 1024-row band of an 8-GPU split).  A unit that streams northwards reads its lower halo FIRST and its upper
// halo LAST; its neighbour above, streaming northwards too, reads those same rows first, a whole unit's time earlier or
// later: by then they have left the L2.  So every other segment of the cut streams the other way: a northward unit and the
// southward unit above it END on the same sixteen rows, that southward unit and the northward unit above IT START on the same
// sixteen rows -- at the same time when they were dispatched together (a band's single round), behind the same L2 (units of
// a strip share an XCD: st_units' XCD-aware order) -- and one of the two finds them in the cache.
// A southward unit is the same program on the mirrored band: the kernel takes a second view of the band (`a_dn`: pointers to
// the last row, a negative row stride, the row tables reversed, y_first and the poles mirrored, a unit's rows counted from
// the top) and a unit's flag picks the view.  The formula is symmetric in north and south -- (pS + pN) and the land-neighbour
// counts commute -- and the code bits are swapped on arrival like a mirrored row's: the same bits, by construction and by
// tests/stream_worker.py (every case runs with and without DMG_ST_FLIP).
// Measured (profiles/r04_southward_units.txt, one build, DMG_ST_FLIP=0/1, two repetitions each): 16384x8192 survey 3.931 / 3.938 ->
// 3.959 / 3.970 M (+0.8 %), all-ocean 2.933 / 2.943 -> 2.997 / 2.998 M (+2.0 %), 16384x2048 +0.5 / +1.5 %, a 16384x1024 band
// +0.2 % / within its noise (the middle band of an 8-way split with its ghost rows: 59.8 -> 59.3 us per launch).  Small, never a
// loss: the default.  What it says about the small band: its 36 % of re-read halo rows are not what it waits for.
template <int DV>
__global__ __launch_bounds__(64, 2) void k_jacobi_stream(st_args a_up, st_args a_dn)
{
    __shared__ v4f bring[ST_RING][64];
    __shared__ unsigned cring[ST_RING][64];          // coast rows: land-neighbour counts, eight fp4 values per row and lane (st_row_any)
    __shared__ v4f ktab[ST_MAXROWS + 2 * TB_T + 6];   // coefficients {cx, 1/cx, Beta, 1/Beta} of every window row, filled once
    const int lane = threadIdx.x;
    const int4 u = a_up.units[blockIdx.x];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const st_args &a = (__builtin_amdgcn_readfirstlane(u.w) & 2) ? a_dn : a_up;                // bit 1: a unit that streams southwards (its rows are the mirrored view's)
    const int W = a.W;
    if (ST_OPEN_UNITS && !ST_DBG(a, 1) && (__builtin_amdgcn_readfirstlane(u.w) & 1)) {       // an open-water unit (st_units flags it)
        if (a.prof && lane == 0) {
            a.prof[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
            a.prof[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        }
        st_open_unit<DV>(a, bring, ktab, lane, tx, r0, r1);
        if (a.prof && lane == 0) a.prof[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    if (a.prof && lane == 0) {
        a.prof[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        // where the unit ran: HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID
        a.prof[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
    // two window rows per step, and an even number of steps: the loop body is two steps with no branch in between, so that
    // every wait for a staged load is a counted vmcnt (a conditional second step makes the compiler wait for vmcnt(0))
    const int nsteps = (((r1 - r0) + 2 * TB_T + 3) / 4) * 2;
    const int e0 = r0 - TB_T;                       // unclamped local row of the first window row
    int col = tx * TB_OW - TB_T + 4 * lane;         // this lane's 4 columns (wrapped), plain and pole-shifted
    const int xcol = col;                           // unwrapped, for the store
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);

    // (Measured and dropped, round 3: requesting the window's next 8 .. 32 rows up front -- LDS-DMA loads into a scrap of
    // LDS -- so that the pipeline's own requests hit the L2 while it fills.  Bit-identical, and slower: a 16384 x 1024 band's
    // launch went from 67 to 71 / 74 / 80 us.  A unit's fixed cost is not memory latency: with every load pointed at cached
    // rows and the stores dropped the same launch still takes 58 us; see DESIGN.md 4.1b, "what a unit's first steps cost".)
    // the first two steps' rows are requested before anything else: their round trip covers the tables' below
    st_state z;
    {
        const int e_last = e0 + 2 * nsteps + 5;             // the last row a step requests (st_step: e_t + 5)
        z.plain = e0 >= 0 && e_last < a.L && a.y_first + e0 >= 0 && a.y_first + e_last < a.H;
        z.nx_p = a.pin + (long)e0 * a.rs; z.nx_b = a.b + (long)e0 * a.rs; z.nx_c = a.code + (long)e0 * a.rs;
    }
    st_fetch<false>(a, z, 0, 0, e0, col, colp);
    st_fetch<false>(a, z, 0, 1, e0 + 1, col, colp);
    st_fetch<false>(a, z, 1, 0, e0 + 2, col, colp);          // (rows past the window are clamped into the band: harmless)
    st_fetch<false>(a, z, 1, 1, e0 + 3, col, colp);

    // classes of the window rows: rows beyond the grid or the stored band count as mixed (their codes decide)
    unsigned cls_lo = 0x02020202u, cls_hi = 0u;              // (cls_lo: four classes per lane, st_class4)
    if (a.rowclass) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int lu = e0 + lane + 64 * q, yg = a.y_first + lu;
            if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls_lo = (cls_lo & ~(0xFFu << (8 * q))) | ((unsigned)a.rowclass[(size_t)tx * a.L + lu] << (8 * q));
        }
    }
    if (DV && a.rowbad) {                                    // (mirrored and clamped rows have their image's divisors)
#pragma unroll
        for (int q = 0; q < 4; q++) cls_lo |= ((unsigned)a.rowbad[st_map_row(a, e0 + lane + 64 * q).l] & 1u) << (8 * q + 2);
    }

#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0xFFFFFFFFu;                                // rows before the window: "land", so that a level idles until its rows exist
    z.mixbits = 0; z.slot = 0; z.badbits = 0;
    z.fc[0][0] = st_class4(cls_lo, 0); z.fc[0][1] = st_class4(cls_lo, 1); z.fc[1][0] = st_class4(cls_lo, 2); z.fc[1][1] = st_class4(cls_lo, 3);
    {
        // one 16-byte table entry per window row (two spare: the next level's prefetch): all of a lane's (up to four) loads in
        // flight together -- a loop of load, wait, store costs a unit a memory round trip per 64 rows
        const int n = 2 * nsteps + 2;
        v4f kt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) kt[q] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)st_map_row(a, e0 + min(lane + 64 * q, n - 1)).l * 8));
#pragma unroll
        for (int q = 0; q < 4; q++) if (lane + 64 * q < n) ktab[lane + 64 * q] = kt[q];
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    // (the two waves of a SIMD taking turns at priority, as in k_jacobi_stream_twin: measured here, 16384 x 8192, and dropped -- 277.5
    // against 272.4 us, 364.7 against 359.3 without land: in a launch of several rounds the older wave ending early is what
    // staggers the rounds)
    for (int t = 0; t < nsteps; t += 2) {
#ifdef ST_EXPERIMENTS
        if (a.stepprof && lane == 0 && t < 44) a.stepprof[(size_t)blockIdx.x * 64 + t / 2] = __builtin_amdgcn_s_memrealtime();
#endif
        st_step<0, false, DV>(a, z, bring, cring, ktab, lane, t, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol);
        st_step<1, false, DV>(a, z, bring, cring, ktab, lane, t + 1, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol);
    }
    if (a.prof && lane == 0) a.prof[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef ST_EXPERIMENTS
    if (a.stepprof && lane == 0) a.stepprof[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Twin units (st_step's TW form, described there): a workgroup of two waves per pair -- wave 0 the unit above the seam (the
// plain view), wave 1 the unit below it (the mirrored view); a_up.units holds two entries per workgroup.  The window of a twin
// unit starts ON its first row (no halo towards the seam); ST_TWIN_MAXROWS keeps the two waves' tables within a quarter of a
// CU's LDS: four workgroups, eight waves, two per SIMD as before.
template <int DV>
__global__ __launch_bounds__(128, 2) void k_jacobi_stream_twin(st_args a_up, st_args a_dn)
{
    __shared__ v4f bring2[2][ST_RING][64];
    __shared__ unsigned cring2[2][ST_RING][64];
    __shared__ v4f ktab2[2][ST_TWIN_MAXROWS + TB_T + 8];
    // the rows the two waves hand each other travel through ring slots 10 and 11 of the SENDER's right-hand-side ring: a unit's own
    // rows reach those slots with the arrivals of step 5, the last hand-over is read after step 3, and one more barrier at the top of
    // step 4 (st_step) keeps the two apart whatever the waves' pace -- no LDS of their own, which is what lets the tables hold 232 rows
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    v4f (*bring)[64] = bring2[wave];
    unsigned (*cring)[64] = cring2[wave];
    v4f *ktab = ktab2[wave];
#ifdef ST_EXPERIMENTS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int4 u = a_up.units[2 * blockIdx.x + wave];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const int uflags = __builtin_amdgcn_readfirstlane(u.w);
    if (uflags & 8) return;                            // a workgroup that only holds its place in the list (st_place_twins): both its entries say so
    const st_args &a = (uflags & 2) ? a_dn : a_up;     // (either unit of a pair can be the workgroup's first wave)
    const int W = a.W;
    if (a.prof && lane == 0) {
        a.prof[3 * (2 * blockIdx.x + wave)] = __builtin_amdgcn_s_memrealtime();
        a.prof[3 * (2 * blockIdx.x + wave) + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
    // (at least four steps, whatever the unit's rows: the partner's pipeline fills through this wave's first four)
    const int nsteps = (((r1 - r0) + TB_T + 3) / 4) * 2;
    const int e0 = r0;                                // the window starts on the seam
    int col = tx * TB_OW - TB_T + 4 * lane;
    const int xcol = col;
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
    st_state z;
    {
        const int e_last = e0 + 2 * nsteps + 5;
        z.plain = e0 >= 0 && e_last < a.L && a.y_first + e0 >= 0 && a.y_first + e_last < a.H;
        z.nx_p = a.pin + (long)e0 * a.rs; z.nx_b = a.b + (long)e0 * a.rs; z.nx_c = a.code + (long)e0 * a.rs;
    }
    st_fetch<false>(a, z, 0, 0, e0, col, colp);
    st_fetch<false>(a, z, 0, 1, e0 + 1, col, colp);
    st_fetch<false>(a, z, 1, 0, e0 + 2, col, colp);
    st_fetch<false>(a, z, 1, 1, e0 + 3, col, colp);
    unsigned cls_lo = 0x02020202u, cls_hi = 0u;
    if (a.rowclass) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int lu = e0 + lane + 64 * q, yg = a.y_first + lu;
            if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls_lo = (cls_lo & ~(0xFFu << (8 * q))) | ((unsigned)a.rowclass[(size_t)tx * a.L + lu] << (8 * q));
        }
    }
    if (DV && a.rowbad) {
#pragma unroll
        for (int q = 0; q < 4; q++) cls_lo |= ((unsigned)a.rowbad[st_map_row(a, e0 + lane + 64 * q).l] & 1u) << (8 * q + 2);
    }
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0xFFFFFFFFu;
    z.mixbits = 0; z.slot = 0; z.badbits = 0;
    z.fc[0][0] = st_class4(cls_lo, 0); z.fc[0][1] = st_class4(cls_lo, 1); z.fc[1][0] = st_class4(cls_lo, 2); z.fc[1][1] = st_class4(cls_lo, 3);
    {
        const int n = 2 * nsteps + 2;
        v4f kt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) kt[q] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)st_map_row(a, e0 + min(lane + 64 * q, n - 1)).l * 8));
#pragma unroll
        for (int q = 0; q < 4; q++) if (lane + 64 * q < n) ktab[lane + 64 * q] = kt[q];
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    v4f (*xme)[64] = &bring2[wave][ST_RING - 2];
    const v4f (*xpeer)[64] = &bring2[wave ^ 1][ST_RING - 2];
#if ST_TWIN_FAIR
    // The SIMD's arbiter serves the older of its two waves first: of two equal units the older ends after 37 us, the younger after
    // 44, and runs its last stretch alone at two thirds of the SIMD's throughput.  Here the two take turns at a raised priority
    // (s_setprio), iteration by iteration, so that they end together: the 8-way band 48.2 -> 47.7 us per launch, the 4-way band
    // 80.5 -> 78.3 (same-box A/B of two builds; turns every step: the same; every other iteration: nothing)
    const int wslot = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11))) & 1;
#endif
    for (int t = 0; t < nsteps; t += 2) {
#if ST_TWIN_FAIR == 4
        if (wslot) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#elif ST_TWIN_FAIR
        if (uflags & 16) __builtin_amdgcn_s_setprio(3);          // an edge unit: ahead of whatever shares its SIMD, all the way
        else if (((t >> ST_TWIN_FAIR) ^ wslot) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
#ifdef ST_EXPERIMENTS
        if (a.stepprof && lane == 0 && t < 44) a.stepprof[(size_t)(2 * blockIdx.x + wave) * 64 + t / 2] = __builtin_amdgcn_s_memrealtime();
#endif
        st_step<0, false, DV, 1>(a, z, bring, cring, ktab, lane, t, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol, 0, xme, xpeer);
#if ST_TWIN_FAIR == 4
        if (wslot) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
#endif
        st_step<1, false, DV, 1>(a, z, bring, cring, ktab, lane, t + 1, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol, 0, xme, xpeer);
    }
    if (a.prof && lane == 0) a.prof[3 * (2 * blockIdx.x + wave) + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef ST_EXPERIMENTS
    if (a.stepprof && lane == 0) { a.stepprof[(size_t)(2 * blockIdx.x + wave) * 64 + 63] = __builtin_amdgcn_s_memrealtime(); a.stepprof[(size_t)(2 * blockIdx.x + wave) * 64 + 62] = t_entry; }
#endif
}

// =================================================================================================================
// The persistent form: the same wave program, but a wave does not end with its unit.
//
// Measured on k_jacobi_stream (DMG_ST_PROF, one line per unit): identical units take 120 .. 183 us depending on what
// their neighbours on the CU and in the memory system are doing, so however the units are cut, the last of a launch's
// ~2.4 rounds leaves 8-11 % of the wave slots idle; and a unit of n rows pays for n + 15 (the pipeline's fill and drain),
// which at the 84-104 rows a balanced cut allows is another 15 %.  Here the launch has exactly as many waves as the chip
// has slots.  Every wave first takes segments from a queue (cut so that there is about one per wave: ~270 rows on the
// full grid, 5 % fill overhead), and a wave that finds the queue empty takes the upper part of the segment with the most
// rows left: the owner of a segment publishes its progress, re-reads its end row every four rows, and a thief moves the end
// row down with one compare-and-swap.  Nobody ever waits for anybody: a wave that finds nothing worth taking (fewer than
// STP_MIN_STEAL rows left anywhere) ends, so the grid drains by itself.
//
// Why a stale view is harmless: every row a wave stores is the correct value of that row (a level's row is computed
// from three correct rows of the level below, wherever the wave started), so a row stored by two waves is stored twice
// with the same bits.  What must hold is only that every row of a segment is stored by somebody: the thief stores
// [h', end) for the end it swapped out, and the owner stops at no end it has not read from the word itself.
// =================================================================================================================
// A segment's word: strip (12 bits) | first row + STP_BIAS (26 bits) | end row + STP_BIAS (26 bits).  Everything a thief
// needs to stay inside the segment is in this one word; the owner's progress (seg_cur) is advisory -- whatever a thief
// reads there is clamped into [first - 8, end], so a stale or not yet written value only makes the split less even.

__global__ void k_st_reset(unsigned long long *word, unsigned *cur, int *ctl, const unsigned long long *word0,
                                                     const unsigned *cur0, int nseg0, int total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { word[i] = i < nseg0 ? word0[i] : 0ull; cur[i] = i < nseg0 ? cur0[i] : 0u; }
    if (i == 0) { ctl[0] = 0; ctl[1] = nseg0; }
}

__global__ __launch_bounds__(64, 2) void k_jacobi_stream_p(st_args a)
{
    __shared__ v4f bring[ST_RING][64];
    __shared__ unsigned cring[ST_RING][64];
    __shared__ v4f ktab[STP_KT];                     // coefficients of window rows, a ring refilled every 64 rows
    const int lane = threadIdx.x;
    const int W = a.W;
    unsigned scan_from = (blockIdx.x * 2654435761u) >> 8;             // where this wave starts to look for a victim
    if (a.prof && lane == 0) a.prof[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    unsigned prof_rows = 0, prof_q = 0, prof_s = 0;
    unsigned long long prof_look = 0;
    for (;;) {
        // ---- a segment: the next one of the queue, or the upper part of what somebody still has to do ----
        int id, tx, lo, hi;
        {
            int q = 0;
            if (lane == 0) q = atomicAdd(&a.seg_ctl[0], 1);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q < a.nseg0) {
                id = q;
                unsigned long long w = __hip_atomic_load(&a.seg_word[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)w);
                tx = stp_tx(w); lo = stp_lo(w); hi = stp_hi(w);
                prof_q++;
            } else {
                bool got = false;
                if (ST_DBG(a, 16)) break;
                const unsigned long long look0 = a.prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
                const int nwin = a.cap / 512;
                const unsigned start = scan_from >> 8;
                for (int win = 0; win < nwin && !got; win++) {
                    // one window of 512 entries per look (8 per lane, all loads in flight together: a look costs one memory
                    // round trip, a pass over the whole table a handful -- the wave that finds nothing must not take long to
                    // find that out, the launch ends when the last one has).  Thieves start at different windows, and within
                    // a window every thief ranks the candidates of the largest size class (rows left, to a factor of two) in
                    // its own random order: a launch's waves all run dry at about the same time, and with everybody after the
                    // one largest remainder a steal was won by one and retried by dozens (measured: 250 us per wave spent looking)
                    const unsigned base = (start + (unsigned)win) % (unsigned)nwin * 512u;
                    unsigned best = 0;                                   // size class << 24 | random rank << 12 | slot within the window
                    unsigned long long bw = 0ull;
                    int bl = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const unsigned sl = 64u * k + lane, i = base + sl;
                        const unsigned long long w = __hip_atomic_load(&a.seg_word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int c = (int)__hip_atomic_load(&a.seg_cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - STP_BIAS;
                        const int vlo = stp_lo(w), vhi = stp_hi(w);
                        const int lf = vhi - min(max(c, vlo - TB_T), vhi);         // rows left: 0 .. vhi - vlo + 8
                        const unsigned rank = ((sl + 1u) * 2654435761u + scan_from * 40503u) >> 20;
                        const unsigned key = (w != 0ull && vhi > vlo && lf >= STP_MIN_STEAL) ? (((unsigned)(31 - __clz(lf)) << 24) | (rank << 12) | sl) : 0u;
                        if (key > best) { best = key; bw = w; bl = lf; }
                    }
                    unsigned top = best;
#pragma unroll
                    for (int off = 32; off; off >>= 1) top = max(top, (unsigned)__shfl_xor((int)top, off));
                    top = (unsigned)__builtin_amdgcn_readfirstlane((int)top);
                    if (!top) continue;                                  // nothing in this window: the next one
                    scan_from = scan_from * 1664525u + 1013904223u;
                    const unsigned sl = top & 4095u;
                    const int ln = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(best == top));
                    const unsigned long long wv = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bw >> 32), ln) << 32) |
                                                  (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bw, ln);
                    const int left = __builtin_amdgcn_readlane(bl, ln), vhi = stp_hi(wv), vlo = stp_lo(wv), vtx = stp_tx(wv), vcur = vhi - left;
                    // equal shares of the rows still to be taken in: the owner needs [vcur, h' + 8), the thief [h' - 8, vhi + 8)
                    const int hnew = min(max((vhi + vcur) / 2 + 4, vlo + 1), vhi - 1);
                    int ok = 0, slot = 0;
                    if (lane == 0) {
                        unsigned long long expect = wv;
                        ok = __hip_atomic_compare_exchange_strong(&a.seg_word[base + sl], &expect, stp_word(vtx, vlo, hnew), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                        if (ok) {
                            slot = atomicAdd(&a.seg_ctl[1], 1);
                            if (slot >= a.cap) slot = a.cap + (int)blockIdx.x;        // table full: a private entry nobody scans
                            __hip_atomic_store(&a.seg_cur[slot], (unsigned)(hnew - TB_T + STP_BIAS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&a.seg_word[slot], stp_word(vtx, hnew, vhi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    ok = __builtin_amdgcn_readfirstlane(ok);
                    if (!ok) continue;                                   // somebody else was faster (or the owner's end moved): on to the next window
                    id = __builtin_amdgcn_readfirstlane(slot); tx = vtx; lo = hnew; hi = vhi;
                    got = true; prof_s++;
                }
                if (a.prof) prof_look += __builtin_amdgcn_s_memrealtime() - look0;
                if (!got) break;                                         // nothing left worth taking: this wave is done
            }
        }
        if (hi <= lo || lo < 0 || hi > a.L || tx < 0 || tx * TB_OW >= W) continue;      // (cannot happen; never follow a bad entry into memory)

        // ---- stream rows [lo, hi) of strip tx, as k_jacobi_stream does for a unit ----
        const int e0 = lo - TB_T;
        int col = tx * TB_OW - TB_T + 4 * lane;
        const int xcol = col;
        col %= W; if (col < 0) col += W;
        int colp = col + W / 2; if (colp >= W) colp -= W;
        const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
        st_state z;
#pragma unroll
        for (int s = 0; s < 8; s++)
#pragma unroll
            for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int j = 0; j < 2; j++) { z.ps[k][j] = (v4f)(0.0f); z.bs[k][j] = (v4f)(0.0f); z.cs[k][j] = 0; z.ks[k][j] = 2u; }
        z.landbits = 0xFFFFFFFFu;
        z.mixbits = 0; z.slot = 0; z.badbits = 0;
        z.plain = false; z.nx_p = nullptr; z.nx_b = nullptr; z.nx_c = nullptr;
        {
            const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
            z.cy = cf[4]; z.ry = cf[5];
        }
        // window rows [0, 16): the first chapter's refill brings [16, 80)
        if (lane < 16) ktab[lane] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)st_map_row(a, e0 + lane).l * 8));
        st_fetch<true>(a, z, 0, 0, e0, col, colp, tx);
        st_fetch<true>(a, z, 0, 1, e0 + 1, col, colp, tx);
        st_fetch<true>(a, z, 1, 0, e0 + 2, col, colp, tx);
        st_fetch<true>(a, z, 1, 1, e0 + 3, col, colp, tx);
        int t = 0, hi_known = hi;
        for (;;) {
            // a chapter of 64 window rows.  The levels read the coefficients of window rows [2t - 10, 2t + 67) in it: the
            // entries [2t + 16, 2t + 80) replace rows that left the window long ago
            if (!ST_DBG(a, 128) || t == 0)
                ktab[(2 * t + 16 + lane) & (STP_KT - 1)] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)st_map_row(a, e0 + 2 * t + 16 + lane).l * 8));
            const int t_chapter = t + 32;
            int nsteps;
            do {
                // the end row is in the word's low half: a 32-bit load (the unused upper register of a 64-bit one is handed
                // out again at once, and overwriting it means waiting for this load and every staged load before it)
                unsigned w = ~0u;
                if (!ST_DBG(a, 8)) w = __hip_atomic_load(reinterpret_cast<const unsigned *>(&a.seg_word[id]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!ST_DBG(a, 32)) __hip_atomic_store(&a.seg_cur[id], (unsigned)(e0 + 2 * (t + 2) + STP_BIAS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st_step<0, true>(a, z, bring, cring, ktab, lane, t, e0, 0, hi_known, 0u, 0u, col, colp, lane_ok, xcol, tx);
                st_step<1, true>(a, z, bring, cring, ktab, lane, t + 1, e0, 0, hi_known, 0u, 0u, col, colp, lane_ok, xcol, tx);
                t += 2;
                hi_known = min(hi_known, max((int)(((unsigned)__builtin_amdgcn_readfirstlane((int)w)) & STP_MASK) - STP_BIAS, lo));
                nsteps = (((hi_known - lo) + 2 * TB_T + 3) / 4) * 2;
            } while (t < t_chapter && t < nsteps);
            if (t >= nsteps) break;
        }
        // (the owner's progress now stands at or past the end row: nothing is left to take here)
        prof_rows += 2 * t;
    }
    if (a.prof && lane == 0) {
        a.prof[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        a.prof[4 * blockIdx.x + 2] = prof_rows | (prof_look << 32);
        a.prof[4 * blockIdx.x + 3] = ((unsigned long long)prof_s << 32) | prof_q;
    }
}

// =================================================================================================================
// The narrow form: 2 columns per lane (128-column strips, 112 written), everything else as above.  A wave's register
// state halves (8 levels x 4 rows x 2 floats = 64 VGPRs), the kernel fits 128 registers, and FOUR waves share a SIMD
// instead of two.  Measured on the wide form with loads and stores taken out: one wave per SIMD keeps the vector ALU
// 40 % busy, two 68 % -- an in-order wave spends a third of its issue slots on scalar bookkeeping, waits and hazard
// nops, and only other waves can fill them.  The price is a wider relative halo in x (16 of 128 columns instead of 16 of
// 256) and twice the per-visit bookkeeping per cell: +15 % vector and 2.2x scalar instructions (rocprofv3), for a SIMD
// that is then 74 % instead of 70 % busy -- it ends up level with the wide form on the survey terrain and 7 % behind it
// without land, so it is a tuner candidate for small launch geometries, not the default.  The per-row coefficients come through scalar loads from a table padded with the mirrored rows (no LDS table,
// no address VGPRs); right-hand side and codes travel through LDS rings as above.
// =================================================================================================================

struct stn_state {
    v2f w[8][2][2];
    v2f ps[2][2], bs[2][2];
    unsigned cs[2][2];
    unsigned landbits, mixbits;
    int slot;
    float cy, ry;
};
struct stn_coef { float cx, rx, beta, rb; };
typedef const float __attribute__((address_space(4))) *stn_kptr;

// coefficients of window rows y and y + 1: consecutive entries of the padded table, scalar loads
__device__ __forceinline__ void stn_coefs(const st_args &a, int y, stn_coef &ka, stn_coef &kb)
{
    const stn_kptr cf = (stn_kptr)(unsigned long long)(a.jcoef + (size_t)(y + ST_PAD) * 4);
    ka = stn_coef{ cf[0], cf[1], cf[2], cf[3] };
    kb = stn_coef{ cf[4], cf[5], cf[6], cf[7] };
}

__device__ __forceinline__ v2f stn_row_any(const v2f N, const v2f C, const v2f S, const v2f bb, const stn_coef kk, float cy, float ry, bool mixed,
                                           const unsigned short *crow, int lane, int tick)
{
    const jacobi_row k = { kk.cx, cy, kk.beta, kk.rx, ry, kk.rb };
    const float wl = dpp_from_prev_lane(C.y);                // west neighbour of C.x
    const float er = dpp_from_next_lane(C.x);                // east neighbour of C.y
    if (!mixed) {
        const v2f Wp = { wl, C.x }, Ep = { C.y, er };
        return st_value2(N, S, Ep, Wp, bb, k);
    }
    const unsigned c = crow[lane];
    const int t = tick;
    const v2f n0 = { st_sel<0>(c, t, C.x, N.x), st_sel<8>(c, t, C.y, N.y) };
    const v2f s0 = { st_sel<1>(c, t, C.x, S.x), st_sel<9>(c, t, C.y, S.y) };
    const v2f e0 = { st_sel<2>(c, t, C.x, C.y), st_sel<10>(c, t, C.y, er) };
    const v2f w0 = { st_sel<3>(c, t, C.x, wl), st_sel<11>(c, t, C.y, C.x) };
    const v2f r0 = st_value2(n0, s0, e0, w0, bb, k);
    return v2f{ st_zero_land<4>(c, t, r0.x), st_zero_land<12>(c, t, r0.y) };
}

template <int S, int K>
__device__ __forceinline__ void stn_level(const st_args &a, stn_state &z, v2f (*bring)[64], unsigned short (*cring)[64],
                                          int lane, int t, int yA, int r0, int r1, bool lane_ok, int xcol,
                                          const v2f bA, const v2f bB, v2f &bA_next, v2f &bB_next,
                                          const stn_coef kA, const stn_coef kB, stn_coef &kA_next, stn_coef &kB_next)
{
    if (S < 8) {                                              // the next level's operands, while this level computes
        const int sa = st_wrap(z.slot - (S + 1)), sb = sa + 1 == ST_RING ? 0 : sa + 1;
        bA_next = bring[sa][lane]; bB_next = bring[sb][lane];
        stn_coefs(a, yA - 1, kA_next, kB_next);
    }
    // (a level whose rows do not exist yet sees them as land: landbits starts as all ones)
    const unsigned landA = (z.landbits >> (S + 1)) & 1u, landB = (z.landbits >> S) & 1u;
    const unsigned mixA = (z.mixbits >> (S + 1)) & 1u, mixB = (z.mixbits >> S) & 1u;
    const v2f S0 = z.w[S - 1][K ^ 1][0], C0 = z.w[S - 1][K ^ 1][1], N0 = z.w[S - 1][K][0], N1 = z.w[S - 1][K][1];
    v2f oA, oB;
    if (!(landA | landB | mixA | mixB)) {
        const jacobi_row ka = { kA.cx, z.cy, kA.beta, kA.rx, z.ry, kA.rb }, kb = { kB.cx, z.cy, kB.beta, kB.rx, z.ry, kB.rb };
        const float wlA = dpp_from_prev_lane(C0.y), erA = dpp_from_next_lane(C0.x);
        const float wlB = dpp_from_prev_lane(N0.y), erB = dpp_from_next_lane(N0.x);
        const v2f WA = { wlA, C0.x }, EA = { C0.y, erA }, WB = { wlB, N0.x }, EB = { N0.y, erB };
        oA = st_value2(N0, S0, EA, WA, bA, ka);
        oB = st_value2(N1, C0, EB, WB, bB, kb);
    } else {
        const int sa = st_wrap(z.slot - S), sb = sa + 1 == ST_RING ? 0 : sa + 1;
        oA = landA ? (v2f)(0.0f) : stn_row_any(N0, C0, S0, bA, kA, z.cy, z.ry, mixA != 0, cring[sa], lane, t);
        oB = landB ? (v2f)(0.0f) : stn_row_any(N1, N0, C0, bB, kB, z.cy, z.ry, mixB != 0, cring[sb], lane, t);
    }
    if (S < 8) { z.w[S & 7][K][0] = oA; z.w[S & 7][K][1] = oB; }
    else if (lane_ok && !ST_DBG(a, 2)) {
        // only the unit's own rows [r0, r1): in its first steps level 8 runs on rows below r0 (window rows that exist but
        // are not yet valid at this level) -- computed like any halo row, never stored
        if (!landA && yA >= r0 && yA < r1) __builtin_nontemporal_store(oA, reinterpret_cast<v2f *>(a.pout + (size_t)yA * a.W + xcol));
        if (!landB && yA + 1 >= r0 && yA + 1 < r1) __builtin_nontemporal_store(oB, reinterpret_cast<v2f *>(a.pout + (size_t)(yA + 1) * a.W + xcol));
    }
}

__device__ __forceinline__ void stn_fetch(const st_args &a, stn_state &z, int set, int j, int e, int col, int colp)
{
    const st_row m = st_map_row(a, ST_DBG(a, 1) ? (e & 63) : e);
    const size_t off = (size_t)m.l * a.W + (m.shift ? colp : col);
    z.ps[set][j] = *reinterpret_cast<const v2f *>(a.pin + off);
    z.bs[set][j] = *reinterpret_cast<const v2f *>(a.b + off);
    z.cs[set][j] = *reinterpret_cast<const unsigned short *>(a.code + off);
}

template <int K>
__device__ __forceinline__ void stn_arrive(const st_args &a, stn_state &z, v2f (*bring)[64], unsigned short (*cring)[64], int lane,
                                           int j, int e, unsigned cls, int slot, unsigned &land, unsigned &mix)
{
    land = cls == 1u; mix = 0;
    {
        v2f lo;
        asm volatile("v_mov_b64 %0, %1" : "=v"(lo) : "v"(z.ps[K][j]) : "memory");       // a real move (see st_arrive)
        z.w[0][K][j] = lo;
    }
    if (cls == 2u) {
        unsigned c = z.cs[K][j];
        const st_row m = st_map_row(a, e);
        if (m.flip) c = (c & 0xFCFCu) | ((c & 0x0101u) << 1) | ((c & 0x0202u) >> 1);   // mirrored row: swap N/S bits
        const int yg = a.y_first + e;
        if (yg == 0 && !a.spole) c &= ~0x0202u;
        if (yg == a.H - 1 && !a.npole) c &= ~0x0101u;
        mix = __builtin_amdgcn_ballot_w64((c & 0x1F1Fu) != 0) != 0;
        land = __builtin_amdgcn_ballot_w64((c & 0x1010u) != 0x1010u) == 0;
        mix &= ~land;
        if (mix) cring[slot][lane] = (unsigned short)c;
    }
    bring[slot][lane] = z.bs[K][j];
}

template <int K>
__device__ __forceinline__ void stn_step(const st_args &a, stn_state &z, v2f (*bring)[64], unsigned short (*cring)[64], int lane, int t,
                                         int e0, int r0, int r1, unsigned cls_lo, unsigned cls_hi, int col, int colp, bool lane_ok, int xcol)
{
    const int e_t = e0 + 2 * t;
    unsigned landA, mixA, landB, mixB;
    stn_arrive<K>(a, z, bring, cring, lane, 0, e_t, st_class(cls_lo, cls_hi, 2 * t), z.slot, landA, mixA);
    stn_arrive<K>(a, z, bring, cring, lane, 1, e_t + 1, st_class(cls_lo, cls_hi, 2 * t + 1), z.slot + 1, landB, mixB);
    z.landbits = (z.landbits << 2) | (landA << 1) | landB;
    z.mixbits = (z.mixbits << 2) | (mixA << 1) | mixB;
    stn_fetch(a, z, K, 0, e_t + 4, col, colp);
    stn_fetch(a, z, K, 1, e_t + 5, col, colp);
    const int s1 = st_wrap(z.slot - 1);
    v2f b0 = bring[s1][lane], b1 = bring[z.slot][lane], b2, b3;
    stn_coef k0, k1, k2, k3;
    stn_coefs(a, e_t - 1, k0, k1);
    stn_level<1, K>(a, z, bring, cring, lane, t, e_t - 1, r0, r1, lane_ok, xcol, b0, b1, b2, b3, k0, k1, k2, k3);
    stn_level<2, K>(a, z, bring, cring, lane, t, e_t - 2, r0, r1, lane_ok, xcol, b2, b3, b0, b1, k2, k3, k0, k1);
    stn_level<3, K>(a, z, bring, cring, lane, t, e_t - 3, r0, r1, lane_ok, xcol, b0, b1, b2, b3, k0, k1, k2, k3);
    stn_level<4, K>(a, z, bring, cring, lane, t, e_t - 4, r0, r1, lane_ok, xcol, b2, b3, b0, b1, k2, k3, k0, k1);
    stn_level<5, K>(a, z, bring, cring, lane, t, e_t - 5, r0, r1, lane_ok, xcol, b0, b1, b2, b3, k0, k1, k2, k3);
    stn_level<6, K>(a, z, bring, cring, lane, t, e_t - 6, r0, r1, lane_ok, xcol, b2, b3, b0, b1, k2, k3, k0, k1);
    stn_level<7, K>(a, z, bring, cring, lane, t, e_t - 7, r0, r1, lane_ok, xcol, b0, b1, b2, b3, k0, k1, k2, k3);
    stn_level<8, K>(a, z, bring, cring, lane, t, e_t - 8, r0, r1, lane_ok, xcol, b2, b3, b0, b1, k2, k3, k0, k1);
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
}

__global__ __launch_bounds__(64, 4) void k_jacobi_stream_n(st_args a)
{
    __shared__ v2f bring[ST_RING][64];
    __shared__ unsigned short cring[ST_RING][64];
    const int lane = threadIdx.x;
    const int4 u = a.units[blockIdx.x];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const int W = a.W;
    const int nsteps = (((r1 - r0) + 2 * TB_T + 3) / 4) * 2;
    const int e0 = r0 - TB_T;
    int col = tx * STN_OW - TB_T + 2 * lane;
    const int xcol = col;
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    const bool lane_ok = (lane >= TB_T / 2) && (lane < 64 - TB_T / 2) && (xcol < W);
    unsigned cls_lo = 2u, cls_hi = 2u;
    if (a.rowclass) {
        int lu = e0 + lane, yg = a.y_first + lu;
        if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls_lo = a.rowclass[(size_t)tx * a.L + lu];
        lu += 64; yg += 64;
        if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls_hi = a.rowclass[(size_t)tx * a.L + lu];
    }
    stn_state z;
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v2f)(0.0f); z.w[s][k][1] = (v2f)(0.0f); }
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++) { z.ps[k][j] = (v2f)(0.0f); z.bs[k][j] = (v2f)(0.0f); z.cs[k][j] = 0; }
    z.landbits = 0xFFFFFFFFu;                                // rows before the window: "land", so that a level idles until its rows exist
    z.mixbits = 0; z.slot = 0;
    {
        const stn_kptr cf = (stn_kptr)(unsigned long long)a.jcoef_cy;
        z.cy = cf[0]; z.ry = cf[1];
    }
    stn_fetch(a, z, 0, 0, e0, col, colp);
    stn_fetch(a, z, 0, 1, e0 + 1, col, colp);
    stn_fetch(a, z, 1, 0, e0 + 2, col, colp);
    stn_fetch(a, z, 1, 1, e0 + 3, col, colp);
    for (int t = 0; t < nsteps; t += 2) {
        stn_step<0>(a, z, bring, cring, lane, t, e0, r0, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol);
        stn_step<1>(a, z, bring, cring, lane, t + 1, e0, r0, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------
// ---- host ---------------------------------------------------------------------------------------------------------
bool ocean_stream_enabled()
{
    static const bool on = [] { const char *e = getenv("DMG_JACOBI_STREAM"); return !e || atoi(e) != 0; }();
    return on;
}

// the row tables of the mirrored view (k_jacobi_stream's southward units): local row l' of that view is row L - 1 - l'
static int ocean_stream_flipped_tables(dmg_ocean *o)
{
    if (o->jcoef_f) return DMG_OK;
    const int L = o->L;
    const size_t nj = (size_t)L * 8;
    std::vector<float> j(nj), jf(nj);
    auto flip_j = [&](const float *src, float **dst) -> int {
        DMG_HIP(hipMemcpy(j.data(), src, nj * sizeof(float), hipMemcpyDeviceToHost));
        for (int l = 0; l < L; l++) memcpy(&jf[(size_t)l * 8], &j[(size_t)(L - 1 - l) * 8], 8 * sizeof(float));
        DMG_HIP(hipMalloc((void **)dst, nj * sizeof(float)));
        DMG_HIP(hipMemcpy(*dst, jf.data(), nj * sizeof(float), hipMemcpyHostToDevice));
        return DMG_OK;
    };
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    int rc;
    if ((rc = flip_j(o->jcoef, &o->jcoef_f)) || (rc = flip_j(o->jcoef2, &o->jcoef2_f))) return rc;
    {
        std::vector<unsigned char> b((size_t)L, 0), bf((size_t)L);
        DMG_HIP(hipMemcpy(b.data(), o->rowbad, (size_t)L, hipMemcpyDeviceToHost));
        for (int l = 0; l < L; l++) bf[l] = b[L - 1 - l];
        DMG_HIP(hipMalloc((void **)&o->rowbad_f, (size_t)L));
        DMG_HIP(hipMemcpy(o->rowbad_f, bf.data(), (size_t)L, hipMemcpyHostToDevice));
    }
    if (!o->rowclass_host.empty()) {
        const size_t n = o->rowclass_host.size(), tiles_x = n / (size_t)L;
        std::vector<unsigned char> cf(n);
        for (size_t tx = 0; tx < tiles_x; tx++)
            for (int l = 0; l < L; l++) cf[tx * L + l] = o->rowclass_host[tx * L + (L - 1 - l)];
        DMG_HIP(hipMalloc((void **)&o->rowclass_f, n));
        DMG_HIP(hipMemcpy(o->rowclass_f, cf.data(), n, hipMemcpyHostToDevice));
    }
    return DMG_OK;
}

// 8 sweeps over local rows [lo, hi) from p[from] into p[from ^ 1]; seg = rows per unit (0: the default)
int ocean_launch_stream(dmg_ocean *o, int lo, int hi, int part, int zone_lo, int zone_hi, int from, hipStream_t st, int seg, int variant)
{
    if (seg <= 0) seg = st_unit_rows();                    // 0: chosen per geometry
    static const int forced = [] { const char *e = getenv("DMG_ST_VARIANT"); return e ? atoi(e) : -1; }();     // experiments: 0 wide, 1 narrow
    if (variant < 0) variant = forced >= 0 ? forced : 0;   // default: the wide kernel (measured: 2.52 vs 2.51 M on the survey terrain, 1.97 vs 1.83 M without land)
    if (o->grid.W < STN_EW) variant = 0;
    if (variant == 2 && (part != 0 || !o->rowclass32 || o->L + STP_BIAS >= (1 << STP_FIELD) || (o->grid.W + TB_OW - 1) / TB_OW >= 4096)) variant = 0;   // the halves of a split launch keep their unit lists
    if (variant != 2 && seg > (variant == 1 ? STN_MAXROWS : ST_MAXROWS)) seg = variant == 1 ? STN_MAXROWS : ST_MAXROWS;
    const dmg_ocean::tile_list *tl = st_units(o, variant, lo, hi, part, zone_lo, zone_hi, seg);
    if (!tl) {
        if (getenv("DMG_ST_DEBUG")) fprintf(stderr, "[stream] no unit list for rows [%d,%d) part %d: %s\n", lo, hi, part, st_units_failed() ? dmg_last_error() : "allocation failed");
        return st_units_failed() ? DMG_ESTATE : dmg_fail(DMG_ENOMEM, "unit list allocation failed");     // (st_units has said what failed)
    }
    if (tl->n == 0) return DMG_OK;
    st_args a;
    a.pin = o->p[from];
    a.pout = o->p[from ^ 1];
    a.b = o->divw;
    a.code = o->code;
    // the division form: two operations per quotient where every row constant of this band has an exact pair (div2.hip)
    static const bool div2_env = [] { const char *e = getenv("DMG_JACOBI_DIV2"); return !e || atoi(e) != 0; }();
    const bool div2 = variant == 0 && div2_env && o->div2_ok && o->jcoef2;
    a.jcoef = variant == 1 ? o->jcoefp : (div2 ? o->jcoef2 : o->jcoef);
    a.rowbad = div2 && o->div2_nbad ? o->rowbad : nullptr;
    a.jcoef_cy = o->jcoef + 4;                           // row 0's {cy, 1/cy}: the same for every row
    a.W = o->grid.W; a.L = o->L;
    a.rs = o->grid.W; a.flipped = 0;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    static const bool use_cls = getenv("DMG_TB_NOCLASS") == nullptr;
    a.rowclass = use_cls ? (variant == 1 ? o->rowclass_n : o->rowclass) : nullptr;
    a.units = (const int4 *)tl->d;
    a.rowclass32 = o->rowclass32;
    a.seg_word = tl->seg_word; a.seg_cur = tl->seg_cur; a.seg_ctl = tl->seg_ctl; a.nseg0 = tl->n; a.cap = tl->cap;
    o->last_kernel = variant == 2 ? "k_jacobi_stream_p" : (variant ? "k_jacobi_stream_n" : "k_jacobi_stream");
    static const int dbg = [] { const char *e = getenv("DMG_ST_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    a.prof = nullptr;
    a.stepprof = nullptr;
#ifdef ST_EXPERIMENTS
    static const char *stepprof_path = getenv("DMG_ST_STEPPROF");
    static int stepprof_count = 0;
    const bool stepprof_now = stepprof_path && !variant && tl->n_units > 1024 && ++stepprof_count == 40;    // the 40th full-size launch
    if (stepprof_now && (hipMalloc((void **)&a.stepprof, tl->n_units * 64 * 8) != hipSuccess || hipMemset(a.stepprof, 0, tl->n_units * 64 * 8) != hipSuccess)) a.stepprof = nullptr;
#endif
    static const char *prof_path = getenv("DMG_ST_PROF");  // experiments: the first full-size launch writes one line per unit to this file
    static bool prof_done = false;
    static int prof_skip = [] { const char *e = getenv("DMG_ST_PROF_SKIP"); return e ? atoi(e) : 0; }();      // ... the (n+1)-th such launch: a warm one
    const bool prof_now = prof_path && !prof_done && !variant && tl->n_units > 1024 && prof_skip-- <= 0;
    if (prof_now && hipMalloc((void **)&a.prof, tl->n_units * 24) != hipSuccess) a.prof = nullptr;
    static const int extra_lds = [] { const char *e = getenv("DMG_ST_EXTRA_LDS"); return e ? atoi(e) : 0; }();   // experiments: caps the waves per CU
    if (variant == 2) {
        const int total = tl->cap + tl->waves;
        if (prof_path && !prof_done && tl->waves > 1024 && hipMalloc((void **)&a.prof, (size_t)tl->waves * 32) != hipSuccess) a.prof = nullptr;
        hipLaunchKernelGGL(k_st_reset, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tl->seg_word, tl->seg_cur, tl->seg_ctl,
                           (const unsigned long long *)tl->seg_word0, (const unsigned *)tl->seg_cur0, tl->n, total);
        hipLaunchKernelGGL(k_jacobi_stream_p, dim3((unsigned)tl->waves), dim3(64), 0, st, a);
        if (a.prof) {
            prof_done = true;
            std::vector<unsigned long long> tt((size_t)tl->waves * 4);
            DMG_HIP(hipStreamSynchronize(st));
            DMG_HIP(hipMemcpy(tt.data(), a.prof, tt.size() * 8, hipMemcpyDeviceToHost));
            (void)hipFree(a.prof);
            if (FILE *f = fopen(prof_path, "w")) {
                fprintf(f, "wave,t0,t1,rows,queued,stolen,look\n");
                for (int i = 0; i < tl->waves; i++)
                    fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu\n", i, tt[4 * i], tt[4 * i + 1], tt[4 * i + 2] & 0xFFFFFFFFull, tt[4 * i + 3] & 0xFFFFFFFFull,
                            tt[4 * i + 3] >> 32, tt[4 * i + 2] >> 32);
                fclose(f);
            }
            a.prof = nullptr;
        }
    }
    else if (variant) hipLaunchKernelGGL(k_jacobi_stream_n, dim3((unsigned)tl->n), dim3(64), (size_t)extra_lds, st, a);
    else {
        // the mirrored view of the band for the units that stream southwards: pointers to the last stored row, a negative row
        // stride, the row tables reversed (built once per ocean and terrain), grid rows and poles mirrored
        st_args ad = a;
        if (tl->n_flip || tl->twin) {
            int rc = ocean_stream_flipped_tables(o);
            if (rc) return rc;
            const long last = (long)(o->L - 1) * o->grid.W;
            ad.pin = a.pin + last; ad.pout = a.pout + last; ad.b = a.b + last; ad.code = a.code + last;
            ad.rs = -(long)o->grid.W; ad.flipped = 1;
            ad.jcoef = div2 ? o->jcoef2_f : o->jcoef_f;
            ad.rowbad = a.rowbad ? o->rowbad_f : nullptr;
            ad.rowclass = a.rowclass ? o->rowclass_f : nullptr;
            ad.y_first = o->grid.H - a.y_first - o->L;
            ad.spole = a.npole; ad.npole = a.spole;
        }
        if (tl->twin) {
            o->last_kernel = "k_jacobi_stream_twin";
            if (div2) hipLaunchKernelGGL(k_jacobi_stream_twin<1>, dim3((unsigned)tl->n), dim3(128), 0, st, a, ad);
            else hipLaunchKernelGGL(k_jacobi_stream_twin<0>, dim3((unsigned)tl->n), dim3(128), 0, st, a, ad);
        }
        else if (div2) hipLaunchKernelGGL(k_jacobi_stream<1>, dim3((unsigned)tl->n), dim3(64), (size_t)extra_lds, st, a, ad);
        else hipLaunchKernelGGL(k_jacobi_stream<0>, dim3((unsigned)tl->n), dim3(64), (size_t)extra_lds, st, a, ad);
    }
    DMG_HIP(hipGetLastError());
#ifdef ST_EXPERIMENTS
    if (a.stepprof) {
        std::vector<unsigned long long> tt(tl->n_units * 64);
        DMG_HIP(hipStreamSynchronize(st));
        DMG_HIP(hipMemcpy(tt.data(), a.stepprof, tt.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(a.stepprof);
        if (FILE *f = fopen(stepprof_path, "w")) {
            for (int i = 0; i < (int)tl->n_units; i++) { for (int k = 0; k < 64; k++) fprintf(f, "%llu%c", tt[(size_t)i * 64 + k], k == 63 ? '\n' : ','); }
            fclose(f);
        }
    }
#endif
    if (prof_now && a.prof) {
        prof_done = true;
        const int n_units = (int)tl->n_units;
        std::vector<unsigned long long> tt((size_t)n_units * 3);
        std::vector<int> uu((size_t)n_units * 4);
        DMG_HIP(hipStreamSynchronize(st));
        DMG_HIP(hipMemcpy(tt.data(), a.prof, tt.size() * 8, hipMemcpyDeviceToHost));
        DMG_HIP(hipMemcpy(uu.data(), tl->d, uu.size() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(a.prof);
        if (FILE *f = fopen(prof_path, "w")) {
            // (r0, r1: the unit's rows in the band's own numbering; open / mixed / land: the classes of its window's rows; then what its
            // steps are made of, st_unit_features)
            fprintf(f, "tx,r0,r1,open,mixed,land,t0,t1,hwid,xcc,flip,twin,fast,coast,tested,t_open,t_mixed,t_land,f_open,f_mixed,f_land\n");
            for (int i = 0; i < n_units; i++) {
                int cnt[3] = { 0, 0, 0 };
                const unsigned char *cls = o->rowclass_host.empty() ? nullptr : o->rowclass_host.data() + (size_t)uu[4 * i] * o->L;
                const bool flip = (uu[4 * i + 3] & 2) != 0, twin = (uu[4 * i + 3] & 4) != 0;
                const int v0 = uu[4 * i + 1], v1 = uu[4 * i + 2];                   // the unit's rows in its view
                const int b0 = flip ? o->L - v1 : v0, b1 = flip ? o->L - v0 : v1;
                const int w0 = (twin && !flip) ? b0 : b0 - TB_T, w1 = (twin && flip) ? b1 : b1 + TB_T;
                for (int r = w0; r < w1; r++) cnt[(cls && r >= 0 && r < o->L) ? cls[r] : 2]++;
                const st_features ft = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, v0, v1, flip, twin);
                fprintf(f, "%d,%d,%d,%d,%d,%d,%llu,%llu,%llu,%llu,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d\n", uu[4 * i], b0, b1, cnt[0], cnt[2], cnt[1], tt[3 * i], tt[3 * i + 1],
                        tt[3 * i + 2] & 0xFFFFFFFFull, tt[3 * i + 2] >> 32, (int)flip, (int)twin, ft.fast, ft.coast, ft.tested, ft.t_open, ft.t_mixed, ft.t_land,
                        ft.f_open, ft.f_mixed, ft.f_land);
            }
            fclose(f);
        }
    }
    return DMG_OK;
}
