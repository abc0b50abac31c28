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
// The prototype covers open water only and was not widened.
#ifdef DMG_EXPERIMENTS
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
