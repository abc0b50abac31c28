// EXPERIMENT (round 5; built only with `make EXTRA=-DDMG_EXPERIMENTS`; tools/deep_probe.py drives it; profiles/r05_deep_form.txt
// holds what it measured).  The deep form of the streaming Jacobi kernel: SIXTEEN sweeps per trip through HBM.
//
// Round 4 read the streaming kernel (ocean_jacobi_stream.hip: 8 sweeps per launch, one wave per unit) as bound by its bytes.
// Eight levels of four rows are all the registers one wave has, so a deeper pipeline is TWO waves: the FRONT wave of a workgroup
// is the streaming kernel's unit as it was -- it requests the rows, takes them in, runs sweeps 1..8 -- except that its level 8
// hands its pair of rows to LDS instead of memory; the BACK wave takes that pair as ITS level 0, runs sweeps 9..16 and stores.
// The right-hand sides the front wave parked in its LDS ring are still there when the back wave needs them: one ring of 24 rows
// instead of 12, read by both.  Each row of pressure and right-hand side crosses HBM once per SIXTEEN sweeps; the price is a strip
// of 224 output columns instead of 240 (the halo in x is 16) and a window of n + 32 rows instead of n + 16.
//
// What it measured (all-ocean 16384x8192, units of 200 rows, same box, per 16 sweeps): two plain launches 733-788 us; this form,
// bit-identical to them, 801 us with one s_barrier per step, 806 us with the two LDS counters below instead (two hand-over slots),
// 885 us with four slots (the larger ring costs a workgroup per CU), 792 us with four pairs per workgroup so that the two waves of a
// pair share a SIMD (DP_WG=4) -- and 783-797 us with every synchronisation instruction switched off at run time (DMG_DEEP_NOSPIN:
// wrong results): the pair does not lose its time waiting.  It is as long as the plain kernel because it issues as many
// instructions per sweep (one arrival instead of two per sixteen levels, against 224 instead of 240 useful columns and a window of
// n + 32 rows) -- HALF THE HBM BYTES AND THE SAME TIME.  The streaming kernel is not waiting for HBM (the one-sweep kernel moves a
// fifth more bytes per second in the same run): it is bound by instruction issue and the stalls of two in-order waves per SIMD
// (vector ALU busy 50–57 %); round 4's reading is revised.  (A build with the synchronisation compiled OUT ran 516-526 us and looked like proof of the
// opposite for an hour: without it the front wave's hand-over stores are dead stores, and the compiler removes the sweeps behind them.)
// The prototype covers open water only (k_jacobi_deep_open, built with -DDMG_EXPERIMENTS).
//
// ROUND 6: the form pays in the TOLERANCE rendering (DMG_QUIRK_JACOBI_TOLERANCE: 22 instead of 42 formula instructions per level
// visit) -- there the 8-sweep kernel runs into its bytes (1.32 GB in 250 us = 5.3 TB/s), and half the bytes per sweep are +25 % on an
// all-ocean grid (profiles/r06_deep_tolerance.txt).  k_jacobi_deep below is the form widened to everything a streaming unit handles
// (land, coast, poles, band edges), always built; the launcher uses it for the 16-sweep launches of a solve in the tolerance
// rendering (ocean_launch_deep; DMG_JACOBI_DEEP=0 switches it off, =1 uses it in the exact rendering too, where it gains nothing).
#ifndef DP_D
#define DP_D 2                        // hand-over slots (pairs of rows); the shared ring of right-hand sides needs 2 * DP_D + 20 rows
#endif
#define ST_RING (2 * DP_D + 20)
#include "ocean_stream_dev.h"
#include "ocean_stream_plan.h"

#define DP_T 16                       // halo depth = sweeps per launch
#define DP_OW (TB_EW - 2 * DP_T)      // 224 output columns per strip
#define DP_LAG 9                      // the back wave's step t runs beside the front wave's step t + 9

#define DP_BARRIER() do { if (SYNC == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)      // (LDS only: the global loads stay in flight; SYNC 0: timing experiments, wrong results)

// Hand-over without a barrier per step (round 5: in lockstep -- one s_barrier per step -- the pair ran at the pace of a plain unit and
// the halved bytes bought nothing: 801 against 786 us per 16 sweeps all-ocean; with no synchronisation at all, as a timing experiment,
// 526 us).  Two counters in LDS: `prod` = pairs of rows the front wave has put into the two hand-over slots, `cons` = pairs the back
// wave has taken out.  The front wave writes pair p into slot p & 1 once cons >= p - 1, waits for its own LDS writes and publishes
// prod = p + 1 (one lane); the back wave takes pair t once prod > t and says so at once (cons = t + 1).  That also bounds the front
// wave's lead to ten steps, which is what the shared 24-row ring of right-hand sides allows (the row it parks at step t lands on the
// slot of the row the back wave reads last at step t - 12).  A wave never spins for ever: after DP_SPIN_MAX looks it goes on (the
// results are then wrong, the grid drains, and the launch's error word says so).
#define DP_SPIN_MAX (1 << 22)
struct dp_sync { int prod, cons, err, pad; };
__constant__ int dp_nospin = 0;
__device__ __forceinline__ void dp_wait_at_least(const int *word, int want, int *err, unsigned long long *acc = nullptr)
{
    const unsigned long long t0 = acc ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int spins = 0;
    if (dp_nospin & 4) return;                       // experiments (wrong results): not even a look
    if (dp_nospin) { if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) *err = 2; return; }   // ... look once, never wait
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > DP_SPIN_MAX) { *err = 1; break; }
    }
    if (acc) *acc += __builtin_amdgcn_s_memrealtime() - t0;
}
__device__ __forceinline__ void dp_publish(int *word, int value, int lane)
{
    if (!(dp_nospin & 2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes (reads) of the pair are done; the global loads stay in flight
    if (dp_nospin & 8) return;                       // experiments: no flag store either
    if (lane == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifdef DMG_EXPERIMENTS
// ---- open water, inside the grid and the stored band (the prototype: what the form is worth where nothing else interferes) ----
template <int K, int DV, int PHASE, int SYNC>      // PHASE 0: the pipeline fills (level S starts at step S); 1: steady
__device__ __forceinline__ void dp_front_step(const st_args &a, st_state &z, v4f (*bring)[64], const v4f *ktab, v4f (*xout)[64], int lane, int t, int e_t,
                                              unsigned col4, unsigned &rowb, const char *pb, const char *bb, dp_sync *sy, unsigned long long *wacc = nullptr)
{
#pragma unroll
    for (int j = 0; j < 2; j++) {
        v2f lo, hi;
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(lo) : "v"(z.ps[K][j].xy), "v"(z.ps[K][j].zw) : "memory");
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(hi) : "v"(z.ps[K][j].xy), "v"(z.ps[K][j].zw) : "memory");
        z.w[0][K][j] = v4f{ lo.x, lo.y, hi.x, hi.y };
        if (DV == 3) {                               // (tolerance rendering: -c bC in the ring, st_arrive)
            const v4f kc = ktab[2 * t + j];
            const float negc = -kc.w;
            v4f bq = z.bs[K][j]; bq.xy = bq.xy * negc; bq.zw = bq.zw * negc;
            st_park_pairs(&bring[z.slot + j][lane], bq);
        }
        else st_park_pairs(&bring[z.slot + j][lane], z.bs[K][j]);
    }
    const unsigned voA = rowb + col4, voB = voA + (unsigned)a.W * 4u;
    z.ps[K][0] = *reinterpret_cast<const v4f *>(pb + voA); z.bs[K][0] = *reinterpret_cast<const v4f *>(bb + voA);
    z.ps[K][1] = *reinterpret_cast<const v4f *>(pb + voB); z.bs[K][1] = *reinterpret_cast<const v4f *>(bb + voB);
    rowb += (unsigned)a.W * 8u;
    const int s1 = st_wrap(z.slot - 1);
    v4f bq[9], kq[9];
    int kb = 0;
    asm volatile("v_mov_b32 %0, %1" : "=v"(kb) : "s"((2 * t - 8) * 16));
    bq[1] = bring[s1][lane]; bq[0] = bring[z.slot][lane];
    kq[1] = ktab[max(2 * t - 1, 0)]; kq[0] = ktab[max(2 * t - 1, 0) + 1];
    unsigned (*nocring)[64] = nullptr;
    if (PHASE != 0 || 1 <= t) st_level<1, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 1, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0 || 2 <= t) st_level<2, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 2, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0 || 3 <= t) st_level<3, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 3, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0 || 4 <= t) st_level<4, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 4, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0 || 5 <= t) st_level<5, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 5, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0 || 6 <= t) st_level<6, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 6, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0 || 7 <= t) st_level<7, K, 1, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 7, 0, 0, false, 0, bq, kq, kb);
    if (PHASE != 0) {
        const int p = t - TB_T;                      // the pair this step hands over
        if (SYNC == 2) dp_wait_at_least(&sy->cons, p - (DP_D - 1), &sy->err, wacc);
        st_level<8, K, 4, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 8, 0, 0, false, 0, bq, kq, kb, nullptr, 0, 0, xout);
        if (SYNC == 2) dp_publish(&sy->prod, p + 1, lane);
    }
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
    DP_BARRIER();
}

template <int K, int DV, int PHASE, int SYNC>      // PHASE 0: fill; 1: steady (both level-16 rows are rows the unit writes); 2: the last steps (range-tested)
__device__ __forceinline__ void dp_back_step(const st_args &a, st_state &z, v4f (*bring)[64], const v4f *ktab, const v4f (*xin)[64], int lane, int t, int e_t,
                                             int r0, int r1, bool lane_ok, int xcol, unsigned col4, unsigned &rowb, const char *sb, dp_sync *sy, unsigned long long *wacc = nullptr)
{
    if (SYNC == 2) dp_wait_at_least(&sy->prod, t + 1, &sy->err, wacc);
    z.w[0][K][0] = xin[0][lane]; z.w[0][K][1] = xin[1][lane];      // the front wave's level 8, in pair order already
    if (SYNC == 2) dp_publish(&sy->cons, t + 1, lane);
    const unsigned voA = rowb + col4, voB = voA + (unsigned)a.W * 4u;
    rowb += (unsigned)a.W * 8u;
    const int s1 = st_wrap(z.slot - 1);
    v4f bq[9], kq[9];
    int kb = 0;
    asm volatile("v_mov_b32 %0, %1" : "=v"(kb) : "s"((2 * t - 8) * 16));
    bq[1] = bring[s1][lane]; bq[0] = bring[z.slot][lane];
    kq[1] = ktab[max(2 * t - 1, 0)]; kq[0] = ktab[max(2 * t - 1, 0) + 1];
    unsigned (*nocring)[64] = nullptr;
    constexpr int M = PHASE == 1 ? 3 : 1;
    if (PHASE != 0 || 1 <= t) st_level<1, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 2 <= t) st_level<2, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 3 <= t) st_level<3, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 4 <= t) st_level<4, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 5 <= t) st_level<5, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 6 <= t) st_level<6, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 7 <= t) st_level<7, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
    // (voA, voB address rows e_t + 4, e_t + 5; this wave's level 8 makes rows e_t - 8, e_t - 7: sb = pout - 12 rows)
    if (PHASE != 0) st_level<8, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb, sb, voA, voB);
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
    DP_BARRIER();
}

// DP_WG pairs per workgroup.  1: a workgroup is one pair (its two waves land on different SIMDs, each beside a wave of another
// pair).  4: a workgroup is a whole CU's worth, waves w and w + 4 are a pair -- the CU deals a workgroup's waves round its four
// SIMDs, so the two waves of a pair SHARE a SIMD and whichever of them waits leaves the SIMD to the other.
#ifndef DP_WG
#define DP_WG 1
#endif
template <int DV, int SYNC>
__global__ __launch_bounds__(128 * DP_WG, DP_WG == 1 ? 2 : 1) void k_jacobi_deep_open(st_args a)
{
    __shared__ v4f bring_[DP_WG][ST_RING][64];
    __shared__ v4f xring_[DP_WG][DP_D][2][64];        // [pair mod DP_D][row of the pair][lane]
    __shared__ v4f ktab_[DP_WG][ST_MAXROWS + 2 * DP_T + 8];   // the FRONT wave's window; the back wave's starts eight entries in
    __shared__ dp_sync sync_[DP_WG];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pr = DP_WG == 1 ? 0 : (wv & (DP_WG - 1));
    const int wave = DP_WG == 1 ? wv : (wv / DP_WG);   // 0: front, 1: back
    v4f (*bring)[64] = bring_[pr];
    v4f (*xring)[2][64] = xring_[pr];
    v4f *ktab = ktab_[pr];
    dp_sync &sync = sync_[pr];
    const int lane = threadIdx.x & 63;
    if (wave == 0 && lane == 0) { sync.prod = 0; sync.cons = 0; sync.err = 0; }
    const int ui = (int)blockIdx.x * DP_WG + pr;
    if (DP_WG > 1 && ui >= a.nseg0) {                 // a pair without a unit (the list's last workgroup): the one barrier, then out
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    const int4 u = a.units[ui];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const int W = a.W;
    const int nsb = (((r1 - r0) + 2 * TB_T + 3) / 4) * 2;        // the back wave's steps: its window is rows [r0 - 8, r1 + 8)
    const int nsf = nsb + TB_T;                                    // the front wave's: rows [r0 - 16, ...): eight steps earlier
    const int e0f = r0 - DP_T, e0b = r0 - TB_T;
    int col = tx * DP_OW - DP_T + 4 * lane;
    const int xcol = col;
    col %= W; if (col < 0) col += W;
    const unsigned col4 = (unsigned)col * 4u;
    unsigned long long waited = 0, *wacc = a.prof ? &waited : nullptr;
    const unsigned long long t_begin = a.prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
    st_state z;
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0; z.mixbits = 0; z.badbits = 0; z.plain = true;
    {
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    if (wave == 0) {
        const char *pb = reinterpret_cast<const char *>(a.pin), *bb = reinterpret_cast<const char *>(a.b);
        unsigned rowb = (unsigned)e0f * (unsigned)W * 4u;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned voA = rowb + col4, voB = voA + (unsigned)W * 4u;
            z.ps[k][0] = *reinterpret_cast<const v4f *>(pb + voA); z.bs[k][0] = *reinterpret_cast<const v4f *>(bb + voA);
            z.ps[k][1] = *reinterpret_cast<const v4f *>(pb + voB); z.bs[k][1] = *reinterpret_cast<const v4f *>(bb + voB);
            rowb += (unsigned)W * 8u;
        }
        z.slot = 0;
        {
            const int n = 2 * nsf + 2;
            v4f kt[5];
#pragma unroll
            for (int q = 0; q < 5; q++) kt[q] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)(e0f + min(lane + 64 * q, n - 1)) * 8));
#pragma unroll
            for (int q = 0; q < 5; q++) if (lane + 64 * q < n) ktab[lane + 64 * q] = kt[q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the table and the counters are there for both waves
        int t = 0;
        for (; t < TB_T; t += 2) {
            dp_front_step<0, DV, 0, SYNC>(a, z, bring, ktab, xring[(t - TB_T) & (DP_D - 1)], lane, t, e0f + 2 * t, col4, rowb, pb, bb, &sync, wacc);
            dp_front_step<1, DV, 0, SYNC>(a, z, bring, ktab, xring[(t + 1 - TB_T) & (DP_D - 1)], lane, t + 1, e0f + 2 * t + 2, col4, rowb, pb, bb, &sync, wacc);
        }
        for (; t < nsf; t += 2) {
            dp_front_step<0, DV, 1, SYNC>(a, z, bring, ktab, xring[(t - TB_T) & (DP_D - 1)], lane, t, e0f + 2 * t, col4, rowb, pb, bb, &sync, wacc);
            dp_front_step<1, DV, 1, SYNC>(a, z, bring, ktab, xring[(t + 1 - TB_T) & (DP_D - 1)], lane, t + 1, e0f + 2 * t + 2, col4, rowb, pb, bb, &sync, wacc);
        }
        for (; t < nsb + DP_LAG; t++) DP_BARRIER();  // the back wave's last steps
    } else {
        const bool lane_ok = (lane >= DP_T / 4) && (lane < 64 - DP_T / 4) && (xcol < W);
        const char *sb = reinterpret_cast<const char *>(a.pout) - (size_t)12 * W * 4;
        unsigned rowb = (unsigned)(e0b + 4) * (unsigned)W * 4u;
        z.slot = TB_T;                                // row e0b sits in the ring slot the front wave gave it: eight rows into its window
        const v4f *ktb = ktab + TB_T;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the table, the counters
        for (int w = 0; w < DP_LAG; w++) DP_BARRIER();           // the front wave's first nine steps: its level 8 has its first rows after the ninth
        // (wall step w = t + DP_LAG; the front wave wrote the pair this step takes into xring[(w - 1) & 1] = xring[t & 1])
        int t = 0;
        for (; t < TB_T; t += 2) {
            dp_back_step<0, DV, 0, SYNC>(a, z, bring, ktb, xring[t & (DP_D - 1)], lane, t, e0b + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, sb, &sync, wacc);
            dp_back_step<1, DV, 0, SYNC>(a, z, bring, ktb, xring[(t + 1) & (DP_D - 1)], lane, t + 1, e0b + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, sb, &sync, wacc);
        }
        for (; t + 2 <= nsb && e0b + 2 * t - 5 < r1; t += 2) {
            dp_back_step<0, DV, 1, SYNC>(a, z, bring, ktb, xring[t & (DP_D - 1)], lane, t, e0b + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, sb, &sync, wacc);
            dp_back_step<1, DV, 1, SYNC>(a, z, bring, ktb, xring[(t + 1) & (DP_D - 1)], lane, t + 1, e0b + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, sb, &sync, wacc);
        }
        for (; t < nsb; t += 2) {
            dp_back_step<0, DV, 2, SYNC>(a, z, bring, ktb, xring[t & (DP_D - 1)], lane, t, e0b + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, sb, &sync, wacc);
            dp_back_step<1, DV, 2, SYNC>(a, z, bring, ktb, xring[(t + 1) & (DP_D - 1)], lane, t + 1, e0b + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, sb, &sync, wacc);
        }
    }
    if (a.prof && lane == 0) {                       // experiments: {start, end, clocks spent waiting for the partner, hardware id} per wave
        unsigned long long *q = a.prof + 4 * (2 * (size_t)ui + wave);
        q[0] = t_begin; q[1] = __builtin_amdgcn_s_memrealtime(); q[2] = waited;
        q[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
}

#endif  // DMG_EXPERIMENTS (the open-water prototype's kernel)

// ---- the generic form (round 6): land, coast, poles, band edges -- everything the streaming kernel's units handle ---------------------
// The FRONT wave is a streaming unit as st_step runs it -- classes, arrival, counts of land neighbours, the three forms of a step --
// whose level 8 goes to the hand-over slots; it also leaves each row's land / coast flags in a small LDS ring.  The BACK wave takes
// the pair as its level 0, builds its own histories from those flags and runs a unit's step without arrival or requests.  Both
// read the right-hand sides and the counts from the front wave's rings (24 rows each).  Same operations on the same values as
// sixteen plain sweeps: bit-identical (tests/test_gpu_jacobi.py::test_deep_form_*).
#define DP_FRING 32
template <int K, int DV, int SYNC>
__device__ __forceinline__ void dp_front_gstep(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], int *fring, const v4f *ktab, v4f (*xring)[2][64],
                                               int lane, int t, int e0, unsigned cls_lo, int col, int colp, int tx, dp_sync *sy)
{
    const int e_t = e0 + 2 * t;
    const int slotB = z.slot + 1;
    unsigned landA, mixA, landB, mixB;
    const unsigned clsA = z.fc[K][0], clsB = z.fc[K][1];
    const float ncA = DV == 3 ? -reinterpret_cast<const float *>(&ktab[2 * t])[3] : 0.0f;
    const float ncB = DV == 3 ? -reinterpret_cast<const float *>(&ktab[2 * t + 1])[3] : 0.0f;
    st_arrive<K, DV>(a, z, bring, cring, lane, 0, e_t, clsA, z.slot, landA, mixA, ncA);
    st_arrive<K, DV>(a, z, bring, cring, lane, 1, e_t + 1, clsB, slotB, landB, mixB, ncB);
    if (lane == 0) { fring[(2 * t) & (DP_FRING - 1)] = (int)(landA | (mixA << 1)); fring[(2 * t + 1) & (DP_FRING - 1)] = (int)(landB | (mixB << 1)); }
    z.landbits = (z.landbits << 2) | (landA << 1) | landB;
    z.mixbits = (z.mixbits << 2) | (mixA << 1) | mixB;
    z.fc[K][0] = st_class4(cls_lo, 2 * t + 4); z.fc[K][1] = st_class4(cls_lo, 2 * t + 5);
    st_fetch(a, z, K, 0, e_t + 4, col, colp, tx, z.fc[K][0]);
    st_fetch(a, z, K, 1, e_t + 5, col, colp, tx, z.fc[K][1]);
    const int s1 = st_wrap(z.slot - 1), r1w = max(2 * t - 1, 0);
    v4f bq[9], kq[9];
    bq[1] = bring[s1][lane]; bq[0] = bring[z.slot][lane]; kq[1] = ktab[r1w]; kq[0] = ktab[r1w + 1];
    int kb = 0;
    asm volatile("v_mov_b32 %0, %1" : "=v"(kb) : "s"((2 * t - 8) * 16));
    const bool filled = t >= TB_T;
    const int p = t - TB_T;                          // the pair this step hands over (filled steps)
    v4f (*xout)[64] = xring[p & (DP_D - 1)];
#define DP_FRONT_LEVELS(M, M8)                                                                                                             \
        st_level<1, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, 0, 0, false, 0, bq, kq, kb);                                    \
        st_level<2, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, 0, 0, false, 0, bq, kq, kb);                                    \
        st_level<3, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, 0, 0, false, 0, bq, kq, kb);                                    \
        st_level<4, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, 0, 0, false, 0, bq, kq, kb);                                    \
        st_level<5, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, 0, 0, false, 0, bq, kq, kb);                                    \
        st_level<6, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, 0, 0, false, 0, bq, kq, kb);                                    \
        st_level<7, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, 0, 0, false, 0, bq, kq, kb);                                    \
        if (SYNC == 2) dp_wait_at_least(&sy->cons, p - (DP_D - 1), &sy->err);                                                             \
        st_level<8, K, M8, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, 0, 0, false, 0, bq, kq, kb, nullptr, 0, 0, xout);              \
        if (SYNC == 2) dp_publish(&sy->prod, p + 1, lane);
    if (filled && !((z.landbits | z.mixbits) & 0x3FFu)) { DP_FRONT_LEVELS(1, 4) }
    else if (filled && !(z.landbits & 0x3FFu)) { DP_FRONT_LEVELS(2, 6) }
    else if (filled) { DP_FRONT_LEVELS(0, 5) }
    else {
        if (1 <= t) st_level<1, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, 0, 0, false, 0, bq, kq, kb);
        if (2 <= t) st_level<2, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, 0, 0, false, 0, bq, kq, kb);
        if (3 <= t) st_level<3, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, 0, 0, false, 0, bq, kq, kb);
        if (4 <= t) st_level<4, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, 0, 0, false, 0, bq, kq, kb);
        if (5 <= t) st_level<5, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, 0, 0, false, 0, bq, kq, kb);
        if (6 <= t) st_level<6, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, 0, 0, false, 0, bq, kq, kb);
        if (7 <= t) st_level<7, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, 0, 0, false, 0, bq, kq, kb);
    }
#undef DP_FRONT_LEVELS
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
    DP_BARRIER();
}

template <int K, int DV, int SYNC>
__device__ __forceinline__ void dp_back_gstep(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const int *fring, const v4f *ktab, const v4f (*xring)[2][64],
                                              int lane, int t, int e0, int r0, int r1, bool lane_ok, int xcol, dp_sync *sy)
{
    const int e_t = e0 + 2 * t;
    if (SYNC == 2) dp_wait_at_least(&sy->prod, t + 1, &sy->err);
    const v4f (*xin)[64] = xring[t & (DP_D - 1)];
    z.w[0][K][0] = xin[0][lane]; z.w[0][K][1] = xin[1][lane];      // the front wave's level 8, in pair order already
    // the two rows' flags as the front wave found them when they arrived (its window starts eight rows below this one's)
    const int fA = __builtin_amdgcn_readfirstlane(fring[(2 * t + TB_T) & (DP_FRING - 1)]), fB = __builtin_amdgcn_readfirstlane(fring[(2 * t + TB_T + 1) & (DP_FRING - 1)]);
    if (SYNC == 2) dp_publish(&sy->cons, t + 1, lane);
    z.landbits = (z.landbits << 2) | ((unsigned)(fA & 1) << 1) | (unsigned)(fB & 1);
    z.mixbits = (z.mixbits << 2) | ((unsigned)((fA >> 1) & 1) << 1) | (unsigned)((fB >> 1) & 1);
    const int s1 = st_wrap(z.slot - 1), r1w = max(2 * t - 1, 0);
    v4f bq[9], kq[9];
    bq[1] = bring[s1][lane]; bq[0] = bring[z.slot][lane]; kq[1] = ktab[r1w]; kq[0] = ktab[r1w + 1];
    int kb = 0;
    asm volatile("v_mov_b32 %0, %1" : "=v"(kb) : "s"((2 * t - 8) * 16));
    const bool filled = t >= TB_T;
#define DP_BACK_LEVELS(M)                                                                                                                  \
        st_level<1, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<2, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<3, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<4, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<5, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<6, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<7, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);                             \
        st_level<8, K, M, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (filled && !((z.landbits | z.mixbits) & 0x3FFu)) { DP_BACK_LEVELS(1) }
    else if (filled && !(z.landbits & 0x3FFu)) { DP_BACK_LEVELS(2) }
    else {
        if (1 <= t) st_level<1, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (2 <= t) st_level<2, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (3 <= t) st_level<3, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (4 <= t) st_level<4, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (5 <= t) st_level<5, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (6 <= t) st_level<6, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (7 <= t) st_level<7, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (8 <= t) st_level<8, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    }
#undef DP_BACK_LEVELS
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
    DP_BARRIER();
}

// class of every (224-column strip, row): as k_tb_rowclass (ocean_jacobi_tb.hip) for the deep form's strips -- 256 columns read,
// 224 written, a halo of 16
__global__ __launch_bounds__(64) void k_dp_rowclass(const unsigned char *__restrict__ code, unsigned char *__restrict__ cls, int W, int L,
                                                    int y_first, int H, int spole, int npole)
{
    const int lane = threadIdx.x, tx = blockIdx.x, l = blockIdx.y;
    int col = tx * DP_OW - DP_T + 4 * lane;
    col %= W; if (col < 0) col += W;
    unsigned c = *reinterpret_cast<const unsigned *>(code + (size_t)l * W + col);
    int yg = y_first + l;
    if (yg == 0 && !spole) c &= ~0x02020202u;
    if (yg == H - 1 && !npole) c &= ~0x01010101u;
    bool any = __builtin_amdgcn_ballot_w64((c & 0x1F1F1F1Fu) != 0) != 0;
    bool all_land = __builtin_amdgcn_ballot_w64((c & 0x10101010u) != 0x10101010u) == 0;
    if (lane == 0) cls[(size_t)tx * L + l] = all_land ? 1 : (any ? 2 : 0);
}

template <int DV, int SYNC>
__global__ __launch_bounds__(128, 2) void k_jacobi_deep(st_args a)
{
    __shared__ v4f bring[ST_RING][64];
    __shared__ unsigned cring[ST_RING][64];
    __shared__ v4f xring[DP_D][2][64];               // [pair mod DP_D][row of the pair][lane]
    __shared__ v4f ktab[ST_MAXROWS + 2 * DP_T + 8];   // the FRONT wave's window; the back wave's starts eight entries in
    __shared__ int fring[DP_FRING];                   // bit 0: the row is land everywhere, bit 1: it has coast (as the front wave's arrival found it)
    __shared__ dp_sync sync;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // 0: front, 1: back
    const int lane = threadIdx.x & 63;
    if (wave == 0 && lane == 0) { sync.prod = 0; sync.cons = 0; sync.err = 0; }
    const int4 u = a.units[blockIdx.x];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const int W = a.W;
    const int nsb = (((r1 - r0) + 2 * TB_T + 3) / 4) * 2;        // the back wave's steps: its window is rows [r0 - 8, r1 + 8)
    const int nsf = nsb + TB_T;                                    // the front wave's: rows [r0 - 16, ...): eight steps earlier
    const int e0f = r0 - DP_T, e0b = r0 - TB_T;
    int col = tx * DP_OW - DP_T + 4 * lane;
    const int xcol = col;
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    st_state z;
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0xFFFFFFFFu;                                // rows before the window: "land", so that a level idles until its rows exist
    z.mixbits = 0; z.badbits = 0;
    {
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    if (wave == 0) {
        {
            const int e_last = e0f + 2 * nsf + 5;             // the last row a step requests
            z.plain = e0f >= 0 && e_last < a.L && a.y_first + e0f >= 0 && a.y_first + e_last < a.H;
            z.nx_p = a.pin + (long)e0f * a.rs; z.nx_b = a.b + (long)e0f * a.rs; z.nx_c = a.code + (long)e0f * a.rs;
        }
        st_fetch(a, z, 0, 0, e0f, col, colp);
        st_fetch(a, z, 0, 1, e0f + 1, col, colp);
        st_fetch(a, z, 1, 0, e0f + 2, col, colp);
        st_fetch(a, z, 1, 1, e0f + 3, col, colp);
        unsigned cls_lo;
        st_unit_tables<DV>(a, lane, tx, e0f, 2 * nsf + 2, cls_lo, ktab);
        z.slot = 0;
        z.fc[0][0] = st_class4(cls_lo, 0); z.fc[0][1] = st_class4(cls_lo, 1); z.fc[1][0] = st_class4(cls_lo, 2); z.fc[1][1] = st_class4(cls_lo, 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the table and the counters are there for both waves
        for (int t = 0; t < nsf; t += 2) {
            dp_front_gstep<0, DV, SYNC>(a, z, bring, cring, fring, ktab, xring, lane, t, e0f, cls_lo, col, colp, tx, &sync);
            dp_front_gstep<1, DV, SYNC>(a, z, bring, cring, fring, ktab, xring, lane, t + 1, e0f, cls_lo, col, colp, tx, &sync);
        }
        if (SYNC == 1) for (int t = nsf; t < nsb + DP_LAG; t++) DP_BARRIER();
    } else {
        const bool lane_ok = (lane >= DP_T / 4) && (lane < 64 - DP_T / 4) && (xcol < W);
        z.slot = TB_T;                                // row e0b sits in the ring slot the front wave gave it: eight rows into its window
        const v4f *ktb = ktab + TB_T;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the table, the counters
        if (SYNC == 1) for (int w = 0; w < DP_LAG; w++) DP_BARRIER();
        for (int t = 0; t < nsb; t += 2) {
            dp_back_gstep<0, DV, SYNC>(a, z, bring, cring, fring, ktb, xring, lane, t, e0b, r0, r1, lane_ok, xcol, &sync);
            dp_back_gstep<1, DV, SYNC>(a, z, bring, cring, fring, ktb, xring, lane, t + 1, e0b, r0, r1, lane_ok, xcol, &sync);
        }
    }
    if (lane == 0 && sync.err && a.deep_err) atomicOr(a.deep_err, 1u);
}

// ---- host: the generic form ---------------------------------------------------------------------------------------------------
bool ocean_deep_enabled(const dmg_ocean *o)
{
    static const int env = [] { const char *e = dmg_hook("DMG_JACOBI_DEEP"); return e ? atoi(e) : -1; }();     // -1: where it pays (the tolerance rendering)
    if (env == 0) return false;
    const bool tol = (o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE) != 0;
    if (env < 0 && !tol) return false;
    // (a unit of the deep form streams its rows + 32: the bands of a split grid, a round of ~36-row units, lose with it -- the
    // middle band of a 2- / 4- / 8-way split in the tolerance rendering 124.5 / 76.2 / 48.1 us per 8 sweeps against 124.0 / 63.5 / 37.3)
    if (env < 0 && o->L < 6000) return false;
    // (the exact rendering's rows without a two-operation divisor pair would need the per-row division forms in both waves: not built)
    if (!tol && o->div2_nbad) return false;
    return ocean_tb_supported(o) && ocean_stream_enabled() && o->grid.W % 4 == 0;
}

// 16 sweeps over local rows [lo, hi) from p[from] into p[from ^ 1]
int ocean_launch_deep(dmg_ocean *o, int lo, int hi, int from, hipStream_t st)
{
    if (hi <= lo) return DMG_OK;
    const int W = o->grid.W, L = o->L;
    const int tiles = (W + DP_OW - 1) / DP_OW;
    if (!o->rowclass_d) {                                  // classes of the deep form's strips: once per terrain (ocean_tb_free_lists drops them)
        DMG_HIP(hipMalloc((void **)&o->rowclass_d, (size_t)tiles * L));
        hipLaunchKernelGGL(k_dp_rowclass, dim3(tiles, L), dim3(64), 0, o->ctx->stream, o->code, o->rowclass_d, W, L, o->y0 - o->gs, o->grid.H,
                           o->topo.spole ? 1 : 0, o->topo.npole ? 1 : 0);
        DMG_HIP(hipGetLastError());
        o->rowclass_d_host.resize((size_t)tiles * L);
        DMG_HIP(hipMemcpyAsync(o->rowclass_d_host.data(), o->rowclass_d, o->rowclass_d_host.size(), hipMemcpyDeviceToHost, o->ctx->stream));
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    }
    if (!o->deep_err) { DMG_HIP(hipMalloc((void **)&o->deep_err, sizeof(unsigned))); DMG_HIP(hipMemsetAsync(o->deep_err, 0, sizeof(unsigned), st)); }
    static const int seg_env = [] { const char *e = dmg_hook("DMG_DEEP_ROWS"); return e ? atoi(e) : 0; }();     // experiments: a fixed unit height
    auto key = std::make_tuple(lo, hi, -8000 - seg_env, 0, 0, 0);
    auto it = o->tb_lists.find(key);
    if (it == o->tb_lists.end()) {
        // Every strip cut into segments of `seg` rows, each trimmed to its first / last row that is not land everywhere, longest
        // first.  The height is chosen per launch geometry and terrain by simulating the dispatch (as st_units does for the plain
        // units): a pair of waves streams its rows + 32 -- taller units recompute less, but the chip holds 4 pairs per CU and the
        // last round should be full.  (Measured, survey terrain, 5000 sweeps: 80 / 120 / 160 / 200 rows 147.8 / 140.1 / 138.1 / 142.0 ms.)
        struct unit { int tx, r0, r1; float cost; };
        const int slots = std::max(o->ctx->num_cu, 1) * 4;
        auto cut = [&](int seg, std::vector<unit> &us) {
            us.clear();
            for (int tx = 0; tx < tiles; tx++) {
                const unsigned char *cls = o->rowclass_d_host.data() + (size_t)tx * L;
                for (int s0 = lo; s0 < hi; s0 += seg) {
                    int r0 = s0, r1 = std::min(s0 + seg, hi);
                    while (r0 < r1 && cls[r0] == 1) r0++;
                    while (r1 > r0 && cls[r1 - 1] == 1) r1--;
                    if (r0 >= r1) continue;
                    float c = 0.0f;
                    for (int r = r0 - DP_T; r < r1 + DP_T; r++) { const int k = (r >= 0 && r < L) ? cls[r] : 2; c += k == 1 ? 0.1f : (k == 2 ? 1.7f : 1.0f); }
                    us.push_back(unit{ tx, r0, r1, c + 12.0f });
                }
            }
            std::stable_sort(us.begin(), us.end(), [](const unit &x, const unit &y) { return x.cost > y.cost; });
        };
        auto makespan = [&](const std::vector<unit> &us) {
            std::vector<float> heap((size_t)slots, 0.0f);
            auto cmp = [](float x, float y) { return x > y; };
            float span = 0.0f;
            for (const unit &u : us) { std::pop_heap(heap.begin(), heap.end(), cmp); heap.back() += u.cost; span = std::max(span, heap.back()); std::push_heap(heap.begin(), heap.end(), cmp); }
            return span;
        };
        std::vector<unit> us, trial;
        if (seg_env >= 8 && seg_env <= 200) cut(seg_env, us);
        else {
            float best = 1e30f;
            for (int seg = 64; seg <= 200; seg += 8) {      // (a front wave's window is n + 32 rows: 256 classes fit its register)
                cut(seg, trial);
                const float sp = makespan(trial);
                if (sp < best) { best = sp; us.swap(trial); }
            }
        }
        std::vector<int> units;
        units.reserve(us.size() * 4);
        for (const unit &u : us) { units.push_back(u.tx); units.push_back(u.r0); units.push_back(u.r1); units.push_back(0); }
        dmg_ocean::tile_list tl;
        tl.n = (int)us.size();
        tl.n_units = us.size();
        if (tl.n) {
            if (hipMalloc((void **)&tl.d, units.size() * sizeof(int)) != hipSuccess) return dmg_fail(DMG_ENOMEM, "unit list allocation failed");
            DMG_HIP(hipMemcpy(tl.d, units.data(), units.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        it = o->tb_lists.emplace(key, tl).first;
    }
    const dmg_ocean::tile_list &tl = it->second;
    if (!tl.n) return DMG_OK;
    const bool tol = (o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE) != 0;
    st_args a{};
    a.pin = o->p[from]; a.pout = o->p[from ^ 1]; a.b = o->divw; a.code = o->code;
    a.jcoef = tol ? o->jcoef3 : (o->div2_ok && o->jcoef2 ? o->jcoef2 : o->jcoef);
    a.jcoef_cy = o->jcoef + 4;
    a.W = W; a.L = L; a.rs = W; a.flipped = 0;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    a.rowclass = o->rowclass_d;
    a.units = (const int4 *)tl.d;
    a.deep_err = o->deep_err;
    o->last_kernel = "k_jacobi_deep";
    if (tol) hipLaunchKernelGGL((k_jacobi_deep<3, 2>), dim3((unsigned)tl.n), dim3(128), 0, st, a);
    else if (o->div2_ok && o->jcoef2) hipLaunchKernelGGL((k_jacobi_deep<1, 2>), dim3((unsigned)tl.n), dim3(128), 0, st, a);
    else hipLaunchKernelGGL((k_jacobi_deep<0, 2>), dim3((unsigned)tl.n), dim3(128), 0, st, a);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

// after a solve's launches: did a wave of the deep form ever give up waiting for its partner?  (a protocol error, not a data-dependent
// event: the results are then wrong and the call that finds the word set fails)
int ocean_deep_check(dmg_ocean *o, hipStream_t st)
{
    if (!o->deep_err) return DMG_OK;
    unsigned e = 0;
    DMG_HIP(hipMemcpyAsync(&e, o->deep_err, sizeof e, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if (e) { (void)hipMemsetAsync(o->deep_err, 0, sizeof(unsigned), st); return dmg_fail(DMG_ESTATE, "deep Jacobi form: a wave gave up waiting for its partner (protocol error)"); }
    return DMG_OK;
}

#ifdef DMG_EXPERIMENTS
// ---- host -------------------------------------------------------------------------------------------------------------
// 16 sweeps over local rows [lo, hi) from p[from] into p[from ^ 1], every strip cut into units of `seg` rows: the prototype's
// launcher (open water only: the caller vouches for the terrain; rows [lo - 16, hi + 24) must lie inside the grid and the band)
int ocean_launch_deep_open(dmg_ocean *o, int lo, int hi, int seg, int from, hipStream_t st)
{
    const int W = o->grid.W;
    if (seg < 4 || seg > ST_MAXROWS) seg = 200;
    const int y_first = o->y0 - o->gs;
    if (lo - DP_T < 0 || hi + DP_T + 8 > o->L || y_first + lo - DP_T < 0 || y_first + hi + DP_T + 8 > o->grid.H || W < TB_EW || W % 4)
        return dmg_fail(DMG_EINVAL, "deep form (prototype): rows [%d, %d) need 16 rows below and 24 above inside the band and the grid", lo, hi);
    const bool tol = (o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE) != 0;
    const bool div2 = !tol && o->div2_ok && o->jcoef2 && !o->div2_nbad;
    auto key = std::make_tuple(lo, hi, -7000 - seg, 0, 0, 0);
    auto it = o->tb_lists.find(key);
    if (it == o->tb_lists.end()) {
        const int tiles = (W + DP_OW - 1) / DP_OW;
        std::vector<int> units;
        // longest first is moot (all alike); strips of one XCD's share next to each other, row block by row block
        for (int r = lo; r < hi; r += seg)
            for (int tx = 0; tx < tiles; tx++) { units.push_back(tx); units.push_back(r); units.push_back(std::min(r + seg, hi)); units.push_back(0); }
        dmg_ocean::tile_list tl;
        tl.n = (int)(units.size() / 4);
        tl.n_units = (size_t)tl.n;
        if (hipMalloc((void **)&tl.d, units.size() * sizeof(int)) != hipSuccess) return dmg_fail(DMG_ENOMEM, "unit list allocation failed");
        DMG_HIP(hipMemcpy(tl.d, units.data(), units.size() * sizeof(int), hipMemcpyHostToDevice));
        it = o->tb_lists.emplace(key, tl).first;
    }
    const dmg_ocean::tile_list &tl = it->second;
    st_args a{};
    a.pin = o->p[from]; a.pout = o->p[from ^ 1]; a.b = o->divw; a.code = o->code;
    a.jcoef = tol ? o->jcoef3 : div2 ? o->jcoef2 : o->jcoef;
    a.jcoef_cy = o->jcoef + 4;
    a.W = W; a.L = o->L; a.rs = W; a.flipped = 0;
    a.y_first = y_first; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    a.units = (const int4 *)tl.d;
    o->last_kernel = "k_jacobi_deep_open";
    static const char *prof_path = dmg_hook("DMG_DEEP_PROF");      // experiments: per-wave clocks of the first launch with more than 1024 units
    static bool prof_done = false;
    a.prof = nullptr;
    if (prof_path && !prof_done && tl.n > 1024 && hipMalloc((void **)&a.prof, (size_t)tl.n * 2 * 4 * 8) != hipSuccess) a.prof = nullptr;
    static const int sync_env = [] { const char *e = dmg_hook("DMG_DEEP_SYNC"); return e ? atoi(e) : 2; }();      // experiments: 0 none (wrong results), 1 a barrier per step, 2 counters (default)
    a.nseg0 = tl.n;
    static const int nospin = [] { const char *e = dmg_hook("DMG_DEEP_NOSPIN"); return e ? atoi(e) : 0; }();
    if (nospin) { static bool set = false; if (!set) { DMG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dp_nospin), &nospin, sizeof nospin)); set = true; } }
    const dim3 grid((unsigned)((tl.n + DP_WG - 1) / DP_WG)), block(128 * DP_WG);
    if (sync_env == 0) hipLaunchKernelGGL((k_jacobi_deep_open<1, 0>), grid, block, 0, st, a);
    else if (sync_env == 1 && DP_WG == 1) hipLaunchKernelGGL((k_jacobi_deep_open<1, 1>), grid, block, 0, st, a);
    else if (tol) hipLaunchKernelGGL((k_jacobi_deep_open<3, 2>), grid, block, 0, st, a);
    else if (div2) hipLaunchKernelGGL((k_jacobi_deep_open<1, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_jacobi_deep_open<0, 2>), grid, block, 0, st, a);
    DMG_HIP(hipGetLastError());
    if (a.prof) {
        prof_done = true;
        std::vector<unsigned long long> tt((size_t)tl.n * 8);
        DMG_HIP(hipStreamSynchronize(st));
        DMG_HIP(hipMemcpy(tt.data(), a.prof, tt.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(a.prof);
        if (FILE *f = fopen(prof_path, "w")) {
            fprintf(f, "unit,wave,t0,t1,waited,hwid,xcc\n");
            for (int i = 0; i < tl.n; i++) for (int w = 0; w < 2; w++) {
                const unsigned long long *q = &tt[(size_t)(2 * i + w) * 4];
                fprintf(f, "%d,%d,%llu,%llu,%llu,%llu,%llu\n", i, w, q[0], q[1], q[2], q[3] & 0xFFFFFFFFull, q[3] >> 32);
            }
            fclose(f);
        }
    }
    return DMG_OK;
}

extern "C" int dmg_ocean_jacobi_deep_probe(dmg_ocean *o, int lo, int hi, int seg, int src)
{
    DMG_CHECK_ARG(o != nullptr && (src == 0 || src == 1) && lo >= 0 && hi > lo);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_set_terrain must come first");
    DMG_HIP(hipSetDevice(o->ctx->device));
    return ocean_launch_deep_open(o, lo, hi, seg, src, o->ctx->stream);
}

#endif  // DMG_EXPERIMENTS
