// Streaming temporally blocked Jacobi: 8 sweeps of the pressure stencil (reference
// src/filter/OceanCurrents.cpp:450-481) per launch, bit-identical to 8 single-sweep launches, with no workgroup
// barrier and no redundant rows beyond a 7/n trapezoid.
//
// Why: k_jacobi_tb8 keeps a 256 x 64 tile in the registers of 8 waves and pays for it three ways --
// 16 of its 64 rows are halo that is recomputed, the two waves holding the halo rows have less work than the six in
// between (one s_barrier per sweep makes everybody wait for the slowest), and a tile's load phase is only hidden if the
// CU's other workgroup happens to be sweeping.  Here ONE wave streams a 256-column strip northwards, row by row, and
// carries all 8 sweeps along as a software pipeline ("time skewing"):
//
//     step t:  rows e = r0 - 8 + 2t and e + 1 arrive from HBM (level 0; requested two steps earlier)
//              level s (1..8) computes sweep s of rows e - s and e - s + 1 from the four level s-1 rows around them: the
//              pair level s-1 made one step ago and the pair it has made in this very step
//              level 8's rows e - 8, e - 7 are final and are stored
//
// A wave keeps, per level, the previous pair of rows in registers (two pairs per level, the loop body is two steps so
// that every slot is a fixed register: 8 levels x 4 rows x 4 floats = 128 VGPRs), the right-hand side and the coast
// rows' land-neighbour counts of the 12 rows in flight in a private LDS ring, and the per-row coefficients of its whole
// window in an LDS table.  East/west neighbours are adjacent lanes (DPP), north/south are the wave's own registers: no
// LDS exchange, no barrier, every wave has the same work, loads and stores trickle along with the arithmetic.  A work
// unit is a strip segment of up to 232 rows: rows [r0-8, r1+8) are read, and level s is only run where its rows are
// still needed ([r0-8+s, r1+8-s)), so the recomputation is 7/n of the work (n = rows per unit) instead of 16/64.  Units
// whose rows are land everywhere are not dispatched; land rows inside a unit cost next to nothing.
//
// What a level visit costs (DESIGN.md 4.1b has the measurements): 50 packed instructions are the formula itself for
// two rows of 256 cells; everything else was taken out piece by piece -- rows are held as cell pairs (c0,c3),(c1,c2) so
// that pW + pE needs no register shuffling; coast rows go through counts of land neighbours (one packed fma per pair
// instead of eight selects); steps without land or coast, and steps with coast but no all-land row, run their eight
// levels without a test or a branch; one ring read per row and level.  Four forms share the code: units (the default for
// the full grid), twin units (round 4, the default for launches of up to 4400 rows -- the bands of a split grid: two waves
// that start on the two sides of a seam and need no halo between them, k_jacobi_stream_twin), narrow units (2 cells per
// lane, five waves per SIMD) and persistent waves with work stealing.
//
// Canvas topology exactly as in ocean_jacobi_tb.hip: columns wrap, rows beyond a pole are the same rows mirrored and
// shifted by W/2 (offset(), src/Shader.h:86-95), rows beyond a non-pole edge are plain mirrors with the clamp bit
// cleared.  Arithmetic: jacobi_value() of ocean_internal.h, the same operations in the same order as every other Jacobi kernel.
#include "ocean_internal.h"
#include "ocean_tb.h"
#include <algorithm>
#include <chrono>

// The DMG_ST_DBG experiments (st_args::dbg) are compiled in with -DST_EXPERIMENTS only: one scalar test and branch per level
// visit costs the kernel 3 % (measured), so the shipped kernels carry none of them
#ifdef ST_EXPERIMENTS
#define ST_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define ST_DBG(a, bit) false
#endif
#define ST_MAXROWS 232            // output rows per unit of the wide form: 248 window rows' classes fit one register (four bytes per
                                  // lane), and the coefficient table of 232 + 22 rows leaves its LDS at 19424 bytes: 8 waves share 160 KB
#define STN_MAXROWS 104           // narrow form: two class registers
#define ST_RING 12                // ring slots (rows): 10 rows are in flight between level 0 and level 8, two arrive
#ifndef ST_FLIP_DEFAULT
#define ST_FLIP_DEFAULT 1         // every other segment of a cut streams southwards (DMG_ST_FLIP=0/1 overrides)
#endif
#ifndef ST_TWIN_DEFAULT
#define ST_TWIN_DEFAULT 1         // launches of about one round are cut into twin units (DMG_ST_TWIN=0/1 overrides; 2: whatever the simulation says)
#endif
#ifndef ST_TWIN_FAIR
#define ST_TWIN_FAIR 1            // k_jacobi_stream_twin: the SIMD's two waves take turns at priority, every iteration (1; 2, 3: every 2, 4; 4: every step; 0: off)
#endif
#ifndef ST_OPEN_UNITS
#define ST_OPEN_UNITS 1           // units whose whole window is plain open water run st_open_unit (0: every unit generic, round 3's kernel)
#endif

struct st_args {
    const float *pin;
    float *pout;
    const float *b;
    const unsigned char *code;
    const float *jcoef;             // [L][8] = cx, rx, beta, rb, cy, ry, -, -  per local row (r* = correctly rounded reciprocals)
    int W, L;
    long rs;                        // elements from a local row to the next: W -- or -W for the units that stream southwards (below)
    int flipped;                    // 1: this is the mirrored view of the band that the southward units run on
    int y_first, H;                 // grid row of local row 0; grid height
    int spole, npole;
    const unsigned char *rowclass;  // [tiles_x][L]: 0 open ocean, 1 land everywhere, 2 mixed; or null
    const int4 *units;              // {tile column, first output row, end output row, -}
    const unsigned char *rowbad;    // DV = 1: [L] 1 = the row's divisors have no exact two-operation pair (its table entry holds (c, r)); or null
    const float *jcoef_cy;          // {cy, 1/cy}: the same for every row
    // persistent form (k_jacobi_stream_p): segment table and queue, see there
    unsigned long long *seg_word;   // [cap + waves]: stp_word(strip, first row, end row); the end row only ever moves down (compare-and-swap by a thief)
    unsigned *seg_cur;              // [cap + waves]: next window row its owner will take + STP_BIAS (owner's progress, plain stores; advisory)
    int *seg_ctl;                   // [0] next queued segment, [1] next free table slot
    int nseg0, cap;
    const unsigned *rowclass32;     // rowclass, one 32-bit word per entry
    unsigned long long *prof;       // experiments only (DMG_ST_PROF): {start, end} of every unit on the 100 MHz s_memrealtime clock, or null
    unsigned long long *stepprof;   // experiments only (-DST_EXPERIMENTS, DMG_ST_STEPPROF): s_memrealtime at the top of the first 64 steps of every unit
    int dbg;                        // experiments only (DMG_ST_DBG bits): 1 = every unit re-reads 64 rows (cache hits), 2 = no stores, 4 = plain instead of nontemporal stores;
                                    // persistent form: 8 = owners do not re-read their end row, 16 = no stealing, 32 = owners do not publish progress,
                                    // 64 = row classes from the table instead of from the codes, 128 = no coefficient refills (timing only: wrong results)
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct st_row { int l; bool flip, shift; };
__device__ __forceinline__ st_row st_map_row(const st_args &a, int local_unclamped)
{
    st_row m;
    int y = a.y_first + local_unclamped;
    m.flip = false; m.shift = false;
    if (y < 0) { y = -1 - y; m.flip = true; m.shift = a.spole != 0; }
    else if (y >= a.H) { y = 2 * a.H - 1 - y; m.flip = true; m.shift = a.npole != 0; }
    int l = y - a.y_first;
    m.l = min(max(l, 0), a.L - 1);      // rows past the stored band only feed cells that are not written
    return m;
}

// x / c on two cells at once: the Markstein sequence of dmg_div_rc (q = x*r; q' = fma(fma(-c, q, x), r, q)) on a register
// pair -- v_pk_mul_f32 + 2 v_pk_fma_f32 with the row constants as SGPR operands.  Written on explicit 2-vectors so that
// the pairing is (cell, cell+1): the N/S operands are then already register pairs and the E/W pairs cost 2 DPP + 4 moves
// per 4 cells; four independent chains per lane-row interleave without wait states.
//
// DV = 1: the division in two operations.  For a constant c with 1/c = h + l (h = RN(1/c), l = RN(1/c - h)) Brisebarre and
// Muller's u = RN(x * l); q = RN(x * h + u) is the correctly rounded x * (1/c) = RN(x / c) for every x -- for most constants;
// which ones is decided per constant.  div2.hip decides it by trying all 2^23 significands of x on the device when an ocean's
// row tables are built (and tries the neighbouring splits for the ~1 % of constants whose standard split fails for a single
// significand): every row constant of every BASELINE grid has such a pair, so a cell costs 10 packed operations instead of
// 13.  The pair sits where (c, r) sit in the Markstein form: c = h, r = l.  About one constant in a thousand has no exact
// pair at all (two significands that pull l in opposite directions): the table keeps (c, r) for such a row, its class byte
// carries a flag (bit 2), and the steps that have it among their ten rows in flight take the tested path, where that row
// goes through the Markstein form (DV = 2: pixelwidth2.y, the same for every row, still as a pair).
template <int DV = 0>
__device__ __forceinline__ v2f st_div2(v2f x, float c, float r)
{
    if (DV == 1) return __builtin_elementwise_fma(x, (v2f)c, x * r);
    const v2f q = x * r;
    const v2f e = __builtin_elementwise_fma((v2f)(-c), q, x);
    return __builtin_elementwise_fma(e, (v2f)r, q);
}
// fc = ((pW + pE)/pw2.x + (pS + pN)/pw2.y - bC) / Beta  (OceanCurrents.cpp:480), statement order of jacobi_value()
__device__ __forceinline__ v2f st_value2(v2f pN, v2f pS, v2f pE, v2f pW, v2f b, const jacobi_row &k)
{
    const v2f t = st_div2(pW + pE, k.cx, k.rx) + st_div2(pS + pN, k.cy, k.ry);
    return st_div2(t - b, k.beta, k.rb);
}
// the same from the two neighbour sums
template <int DV = 0>
__device__ __forceinline__ v2f st_value2s(v2f we, v2f sn, v2f b, const jacobi_row &k)
{
    const v2f t = st_div2<DV>(we, k.cx, k.rx) + st_div2<(DV != 0)>(sn, k.cy, k.ry);
    return st_div2<DV>(t - b, k.beta, k.rb);
}
// Register layout of a row in the wide kernels: a lane's four cells c0..c3 sit in its two register pairs as (c0, c3) and
// (c1, c2).  With the natural pairing (c0, c1), (c2, c3) the east/west operands of a pair straddle two pairs and cost
// 2 DPP + 4 moves + 2 packed adds per row; here pW + pE of the inner cells is one packed add with the second operand's
// halves swapped ((c0 + c2, c3 + c1): op_sel), and of the outer cells two adds with the DPP lane shift folded in
// (prev lane's c3 + c1, next lane's c0 + c2): 3 instructions instead of 8, 10 of the 73 a level visit had.  North/south
// operands and the right-hand side (stored in the ring in this order) are pairs as they are; rows are put into this
// order when they arrive and back into memory order when level 8 stores them.
struct st_we { v2f outer, inner; };
__device__ __forceinline__ st_we st_we_sums(const v4f C)
{
    st_we r;
    r.outer = v2f{ dpp_from_prev_lane(C.y) + C.z, dpp_from_next_lane(C.x) + C.w };
    r.inner = C.xy + C.wz;
    return r;
}
__device__ __forceinline__ v4f st_to_pairs(const v4f m) { return v4f{ m.x, m.w, m.y, m.z }; }     // memory order -> (c0, c3, c1, c2)
// a row of right-hand sides into its ring slot in pair order: two ds_write2_b32 straight from the registers the load filled
// (as one 16-byte store the row first goes through six moves into four consecutive registers)
__device__ __forceinline__ void st_park_pairs(v4f *dst, const v4f m)
{
    typedef __attribute__((address_space(3))) v4f lds_v4f;
    const unsigned addr = (unsigned)(__SIZE_TYPE__)(lds_v4f *)dst;
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write2_b32 %0, %3, %4 offset0:2 offset1:3" :: "v"(addr), "v"(m.x), "v"(m.w), "v"(m.y), "v"(m.z) : "memory");
}
__device__ __forceinline__ v4f st_to_memory(const v4f q) { return v4f{ q.x, q.z, q.w, q.y }; }
// land neighbour -> centre value (OceanCurrents.cpp:470-473) as pK | (pC & mask): land holds exactly +0.0
template <int BIT>
__device__ __forceinline__ float st_sel(unsigned code, int tick, float pC, float pK)
{
    float o;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o) : "v"(pC), "v"(code_mask<BIT>(code, tick)), "v"(pK));
    return o;
}
template <int BIT>
__device__ __forceinline__ float st_zero_land(unsigned code, int tick, float r)      // land centre -> +0.0 (:481)
{
    float o;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(o) : "v"(code_mask<BIT>(code, tick)), "v"(r));
    return o;
}

struct st_state {
    v4f w[8][2][2];                 // w[s][pair][row]: rows of level s (after s sweeps); the pair a step produces, and the one before
    // staging registers, two steps deep: set [t mod 2] holds the pressure, right-hand side and (raw) codes of the pair
    // that arrives at step t; it is refilled for step t + 2 as soon as step t has taken it (a load has two whole steps,
    // ~4-6 us of the wave's arithmetic, to come back: one step was not enough, the waves sat in s_waitcnt a third of the time)
    v4f ps[2][2], bs[2][2];
    unsigned cs[2][2];
    unsigned ks[2][2];              // persistent form: the row's class travels with its loads
    unsigned fc[2][2];              // unit form: the class byte of the row in flight (looked up when its loads are issued: scalar)
    unsigned landbits, mixbits;     // bit j: row (newest - j) is land everywhere / has coast; newest = second row of this step's pair
    // units whose whole window (and the rows requested past it) lies inside the grid and the stored band fetch their rows
    // through three row pointers that advance by one row per fetch (scalar adds) instead of mapping every row index
    const float *nx_p, *nx_b;
    const unsigned char *nx_c;
    bool plain;
    unsigned badbits;               // DV = 1: ... its divisors have no exact two-operation pair (the table holds (c, r) for it)
    int slot;                       // ring slot of the first row of this step's pair (row offset mod ST_RING)
    float cy, ry;                   // the same for every row (pixelsize().y, src/Shader.h:71-74): two SGPRs
};
#define STP_KT 128                // persistent form: the coefficient table is a ring over the window rows
template <bool P> __device__ __forceinline__ int st_kt(int window_row) { return P ? (window_row & (STP_KT - 1)) : window_row; }

// LDS coefficient entries: {cx, 1/cx, 1/Beta, Beta} -- with 1/Beta in the low half of its register pair the packed multiply
// takes it as it is (from the high half the compiler first copies it into a pair of its own: two moves per level visit)
__device__ __forceinline__ v4f st_kswap(const v4f k) { return v4f{ k.x, k.y, k.w, k.z }; }

__device__ __forceinline__ int st_wrap(int slot) { return slot < 0 ? slot + ST_RING : slot; }

// one row of one level, generic: open ocean or coast (wave-uniform choice); N, C, S = the level below's rows y+1, y, y-1
template <int DV = 0>
__device__ __forceinline__ v4f st_row_any(const v4f N, const v4f C, const v4f S, const v4f bb, const v4f kk, float cy, float ry, bool mixed,
                                          const unsigned *crow, int lane)
{
    const jacobi_row k = { kk.x, cy, kk.w, kk.y, ry, kk.z };
    const st_we we = st_we_sums(C);
    v2f o0, o1;
    if (!mixed) {
        o0 = st_value2s<DV>(we.outer, S.xy + N.xy, bb.xy, k);
        o1 = st_value2s<DV>(we.inner, S.zw + N.zw, bb.zw, k);
    } else {
        // coast row.  Land holds exactly +0.0 at every level, so "a land neighbour reads the centre value"
        // (OceanCurrents.cpp:470-473) is pW' + pE' = (pW + pE) + n * pC with n = the number of land neighbours among the
        // two -- exactly, in one rounding: n = 0 adds a zero, n = 1 leaves pK + pC (the other addend is the land cell's
        // +0), n = 2 leaves 0 + 2 pC.  The counts travel as eight fp4 values in one word (0, 1, 2 and -1 are exact in E2M1;
        // one word per lane-row instead of two of fp8 frees 3 KB of LDS for a taller coefficient table), unpacked two at a
        // time by v_cvt_scalef32_pk_f32_fp4, and one v_pk_fma_f32 per pair replaces eight bit-field selects.
        // A land centre carries the count -1 in the N/S word: L = clamp(n + 1) is 0 there and 1 on water, and
        // x = L * t - b (b is -0.0 on land, st_arrive) is then +0, which the divisions by Beta > 0 keep: land -> +0.0 (:481).
        // (A pressure of -0.0, which no sweep that started from +0.0 produces inside the division window of
        // demiurge_hip.h, would read back as +0.0 from an open neighbour pair here.)
        const unsigned c = crow[lane];                        // byte 0: E/W counts of (c0, c3), 1: of (c1, c2), 2 and 3: N/S counts
        const v2f kwe0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(c, 1.0f, 0), kwe1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(c, 1.0f, 1);
        const v2f ksn0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(c, 1.0f, 2), ksn1 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(c, 1.0f, 3);
        const v2f one = { 1.0f, 1.0f };
        v2f L0, L1;
        asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(L0) : "v"(ksn0), "s"(one));
        asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(L1) : "v"(ksn1), "s"(one));
        const v2f we0 = __builtin_elementwise_fma(kwe0, C.xy, we.outer), we1 = __builtin_elementwise_fma(kwe1, C.zw, we.inner);
        const v2f sn0 = __builtin_elementwise_fma(ksn0, C.xy, S.xy + N.xy), sn1 = __builtin_elementwise_fma(ksn1, C.zw, S.zw + N.zw);
        const v2f t0 = st_div2<DV>(we0, k.cx, k.rx) + st_div2<(DV != 0)>(sn0, k.cy, k.ry), t1 = st_div2<DV>(we1, k.cx, k.rx) + st_div2<(DV != 0)>(sn1, k.cy, k.ry);
        o0 = st_div2<DV>(__builtin_elementwise_fma(L0, t0, -bb.xy), k.beta, k.rb);
        o1 = st_div2<DV>(__builtin_elementwise_fma(L1, t1, -bb.zw), k.beta, k.rb);
    }
    return v4f{ o0.x, o0.y, o1.x, o1.y };
}

// one level: sweep S of rows yA = e0 + 2t - S and yB = yA + 1 from level S-1's rows yA-1 .. yB+1 (the pair made one step ago
// and the pair made in this step); K = t mod 2
// MODE 0: the pair's rows as their classes say (tests and branches); 1: both open water; 2: both through the coast formula;
// 3: both open water and known to be rows the unit writes (st_open_unit): level 8 stores them through a scalar base and the
// lane's byte offsets sA, sB, without a range test
template <int S, int K, bool P, int MODE = 0, int DV = 0>
__device__ __forceinline__ void st_level(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const v4f *ktab,
                                         int lane, int t, int yA, int r0, int r1, bool lane_ok, int xcol,
                                         v4f (&bq)[9], v4f (&kq)[9], int kb = 0, const char *sbase = nullptr, unsigned sA = 0, unsigned sB = 0)
{
    // bq[k], kq[k]: right-hand side and coefficients of ring row (newest - k); this level's rows are k = S and k = S - 1
    const v4f bA = bq[S], bB = bq[S - 1], kA = kq[S], kB = kq[S - 1];
    // the rows the next levels need are read from the rings while this level computes.  Level S + 1 works one row further
    // south: its second row is this level's first (already in registers), only its first row, k = S + 1, is new.
    if (S < 8) {
        // coefficients of row k = S + 1, window row 2t - (S + 1); every lane reads the same 16 bytes: one broadcast.
        // Unit form: kb = byte offset of window row 2t - 8 in a VGPR, set once per step -- the level adds a constant (in the
        // first four steps of a unit the offset is negative: the read lands in the ring in front of the table and is not used)
        if (P) kq[S + 1] = ktab[st_kt<P>(max(2 * t - (S + 1), 0))];
        else kq[S + 1] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(ktab) + kb + (8 - (S + 1)) * 16);
    }
    // right-hand sides: the ring slot of row k = S + 1 is odd for even S (slots advance by two per step), so the slot below it
    // exists without a wrap: an even level reads rows S + 1 and S + 2 off one address, the odd levels after the first none
    if (S == 1) bq[2] = bring[st_wrap(z.slot - 2)][lane];
    else if (S < 8 && !(S & 1)) {
        const v4f *pb = &bring[st_wrap(z.slot - (S + 1)) - 1][lane];
        bq[S + 2] = pb[0];
        bq[S + 1] = pb[64];
    }
    // (a level whose rows do not exist yet sees them as land: landbits starts as all ones; rows that exist but are not
    // yet valid at this level are computed like any halo row)
    const v4f S0 = z.w[S - 1][K ^ 1][0], C0 = z.w[S - 1][K ^ 1][1], N0 = z.w[S - 1][K][0], N1 = z.w[S - 1][K][1];
    v4f oA, oB;
    unsigned landA = 0, landB = 0;
    if (MODE == 2) {
        const int sa = st_wrap(z.slot - S), sb = sa + 1 == ST_RING ? 0 : sa + 1;
        oA = st_row_any<DV>(N0, C0, S0, bA, kA, z.cy, z.ry, true, cring[sa], lane);
        oB = st_row_any<DV>(N1, N0, C0, bB, kB, z.cy, z.ry, true, cring[sb], lane);
    } else if (MODE == 1 || MODE == 3 || !(((z.landbits | z.mixbits | (DV ? z.badbits : 0u)) >> S) & 3u)) {   // (one two-bit field of the step's combined mask: the common case costs one scalar test)
        // both rows open ocean (wave-uniform): one block of 8 independent division chains
        const jacobi_row ka = { kA.x, z.cy, kA.w, kA.y, z.ry, kA.z }, kb = { kB.x, z.cy, kB.w, kB.y, z.ry, kB.z };
        const st_we weA = st_we_sums(C0), weB = st_we_sums(N0);
        const v2f a0 = st_value2s<DV>(weA.outer, S0.xy + N0.xy, bA.xy, ka), a1 = st_value2s<DV>(weA.inner, S0.zw + N0.zw, bA.zw, ka);
        const v2f b0 = st_value2s<DV>(weB.outer, C0.xy + N1.xy, bB.xy, kb), b1 = st_value2s<DV>(weB.inner, C0.zw + N1.zw, bB.zw, kb);
        oA = v4f{ a0.x, a0.y, a1.x, a1.y };
        oB = v4f{ b0.x, b0.y, b1.x, b1.y };
    } else {
        landA = (z.landbits >> (S + 1)) & 1u; landB = (z.landbits >> S) & 1u;
        const unsigned mixA = (z.mixbits >> (S + 1)) & 1u, mixB = (z.mixbits >> S) & 1u;
        const int sa = st_wrap(z.slot - S), sb = sa + 1 == ST_RING ? 0 : sa + 1;
        if (DV && ((z.badbits >> S) & 3u)) {
            // a row whose divisors have no exact pair (one in a thousand): Markstein form for that row
            const unsigned badA = (z.badbits >> (S + 1)) & 1u, badB = (z.badbits >> S) & 1u;
            oA = landA ? (v4f)(0.0f) : (badA ? st_row_any<2>(N0, C0, S0, bA, kA, z.cy, z.ry, mixA != 0, cring[sa], lane)
                                             : st_row_any<DV>(N0, C0, S0, bA, kA, z.cy, z.ry, mixA != 0, cring[sa], lane));
            oB = landB ? (v4f)(0.0f) : (badB ? st_row_any<2>(N1, N0, C0, bB, kB, z.cy, z.ry, mixB != 0, cring[sb], lane)
                                             : st_row_any<DV>(N1, N0, C0, bB, kB, z.cy, z.ry, mixB != 0, cring[sb], lane));
        } else {
        oA = landA ? (v4f)(0.0f) : st_row_any<DV>(N0, C0, S0, bA, kA, z.cy, z.ry, mixA != 0, cring[sa], lane);
        oB = landB ? (v4f)(0.0f) : st_row_any<DV>(N1, N0, C0, bB, kB, z.cy, z.ry, mixB != 0, cring[sb], lane);
        }
    }
    if (S < 8) { z.w[S & 7][K][0] = oA; z.w[S & 7][K][1] = oB; }
    else if (MODE == 3) {
        // (a use outside the branch: left alone, the compiler sinks all of level 8 into the `lane_ok` region, where its
        // neighbour-lane adds no longer fuse with their DPP moves)
        asm volatile("" :: "v"(oA), "v"(oB));
        if (lane_ok && !ST_DBG(a, 2)) {
            __builtin_nontemporal_store(st_to_memory(oA), reinterpret_cast<v4f *>(const_cast<char *>(sbase) + sA));
            __builtin_nontemporal_store(st_to_memory(oB), reinterpret_cast<v4f *>(const_cast<char *>(sbase) + sB));
        }
    }
    else if (lane_ok && !ST_DBG(a, 2)) {                          // never re-read in this launch; land rows hold 0 in both buffers already
        // only the unit's own rows [r0, r1): in its first steps level 8 runs on rows below r0, never stored
        if (!landA && yA >= r0 && yA < r1) __builtin_nontemporal_store(st_to_memory(oA), reinterpret_cast<v4f *>(a.pout + (long)yA * a.rs + xcol));
        if (!landB && yA + 1 >= r0 && yA + 1 < r1) __builtin_nontemporal_store(st_to_memory(oB), reinterpret_cast<v4f *>(a.pout + (long)(yA + 1) * a.rs + xcol));
    }
}

// class of window row i (0..127) from the two class registers (lane i holds rows i and 64 + i)
// wide form: lane l holds the classes of window rows l, 64 + l, 128 + l, 192 + l as four bytes
__device__ __forceinline__ unsigned st_class4(unsigned cls4, int i)
{
    // (i < 256 always: a unit has at most ST_MAXROWS = 232 rows, its window 248, its steps' rows 2 t + 1 <= 251)
    return ((unsigned)__builtin_amdgcn_readlane((int)cls4, i & 63) >> (((i >> 6) & 3) * 8)) & 0xFFu;
}
__device__ __forceinline__ unsigned st_class(unsigned cls_lo, unsigned cls_hi, int i)
{
    if (i > 127) return 2u;                                  // the padding rows of the last step: never used
    return (unsigned)__builtin_amdgcn_readlane((int)(i < 64 ? cls_lo : cls_hi), i & 63);
}

// issue the loads of window row e into staging slot [set][j].  Always the same three loads, whatever the row's class:
// the compiler can then count them, and the wait in front of a step's arrival is a counted vmcnt that leaves the other
// staging set's loads (and the stores) in flight -- with class-dependent loads it falls back to vmcnt(0) and the
// prefetch distance collapses to nothing.  (Land rows inside a live unit are few; their loads are wasted bytes.)
// (Measured and dropped, round 3: keeping the last three steps' requests -- six rows past the window, never used -- on the window's
// last row: 3 % of a 180-row unit's load bytes, 12 % of a 28-row unit's, and 1 % slower at every size: a compare and three pointer
// adjustments per request cost more than rows that mostly come out of the Infinity Cache.)
// (Round 2 measured pointing the loads a row does not need -- everything of a land row, the codes of an open-ocean row -- at an
// already cached row as a loss, 2.58 -> 2.49 M: address arithmetic in VGPR pairs.  With scalar row bases the codes of an open
// row ARE redirected now, below; a land row's loads are not: such rows are few inside a live unit.)
template <bool P>
__device__ __forceinline__ void st_fetch(const st_args &a, st_state &z, int set, int j, int e, int col, int colp, int tx = 0, unsigned cls = 2u)
{
    // Three scalar row bases and one 32-bit byte offset per lane: the loads take the SGPR-base + VGPR-offset form, whichever
    // way the bases were found -- advanced by a row (plain units: rows are fetched in order e0, e0 + 1, ...; a step spends a
    // third of its instructions outside its levels, most of them scalar address arithmetic: per-step clocks)
    // or mapped (mirror beyond a pole or an edge, clamp beyond the stored band).  One set of load instructions for both:
    // written as two branches that each load, the compiler merges them and builds every address in a VGPR pair.
    const char *bp, *bb, *bc;
    bool shift = false;
    int ml = 0;
    if (!P && z.plain && !ST_DBG(a, 1)) {
        bp = reinterpret_cast<const char *>(z.nx_p); bb = reinterpret_cast<const char *>(z.nx_b); bc = reinterpret_cast<const char *>(z.nx_c);
        z.nx_p += a.rs; z.nx_b += a.rs; z.nx_c += a.rs;
    } else {
        const st_row m = st_map_row(a, ST_DBG(a, 1) ? (e & 63) : e);
        const long rowoff = (long)m.l * a.rs;
        bp = reinterpret_cast<const char *>(a.pin + rowoff); bb = reinterpret_cast<const char *>(a.b + rowoff); bc = reinterpret_cast<const char *>(a.code + rowoff);
        shift = m.shift; ml = m.l;
    }
    // an open-water row's codes are never looked at (st_arrive): its load goes to the band's first row, which every wave
    // keeps in the L2 -- the same instruction, a ninth of the row's bytes less from HBM
    if (!P && (cls & 3u) == 0u) bc = reinterpret_cast<const char *>(a.code);
    const unsigned cc = (unsigned)(shift ? colp : col);
    z.ps[set][j] = *reinterpret_cast<const v4f *>(bp + cc * 4u);
    z.bs[set][j] = *reinterpret_cast<const v4f *>(bb + cc * 4u);
    z.cs[set][j] = *reinterpret_cast<const unsigned *>(bc + cc);
    if (P && ST_DBG(a, 64)) z.ks[set][j] = a.rowclass32[(size_t)tx * a.L + ml];          // every lane the same word
}
typedef const float __attribute__((address_space(4))) *st_kptr;

// the row that has arrived in staging slot [K][j]: classify it, take its pressure into level 0, park its right-hand side
// and codes in the rings
template <int K>
__device__ __forceinline__ void st_arrive(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], int lane,
                                          int j, int e, unsigned cls, int slot, unsigned &land, unsigned &mix)
{
    cls &= 3u;                                               // (bit 2: the row's divisors have no exact pair, st_step)
    // (the flags as integer arithmetic on scalars: written as comparisons they come out as lane masks, turned into 0 / 1 in a
    // VGPR and read back with v_readfirstlane)
    land = cls & 1u; mix = 0;                                // classes are 0, 1, 2
    // (a land row's pressure is +0 in memory: k_zero_land.)  A real move, opaque to the register coalescer: merged with the
    // staging registers, level 0's rows stay live past the point where the set is reloaded, the reload gets other
    // registers, and the copies the compiler then puts on the loop's back edge wait for loads that were only just issued
    {
        v2f lo, hi;
        // (volatile + "memory": the scheduler must not hoist the set's reload above these moves either -- it did for one
        // of the two sets, whose p rows then went through copies and an early vmcnt wait on the loop's back edge)
        // (v_pk_mov_b32: one half from each source pair -- the row goes into pair order (c0, c3), (c1, c2) on the way)
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(lo) : "v"(z.ps[K][j].xy), "v"(z.ps[K][j].zw) : "memory");
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(hi) : "v"(z.ps[K][j].xy), "v"(z.ps[K][j].zw) : "memory");
        z.w[0][K][j] = v4f{ lo.x, lo.y, hi.x, hi.y };
    }
    if (cls == 2u) {
        unsigned c = z.cs[K][j];
        const st_row m = st_map_row(a, e);
        // mirrored row: swap N/S bits -- and so for every row of a unit that streams southwards (its north is the band's south)
        if (m.flip != (a.flipped != 0)) c = (c & 0xFCFCFCFCu) | ((c & 0x01010101u) << 1) | ((c & 0x02020202u) >> 1);
        // a non-pole edge row's clamp bit only says "the tap lands on the cell itself"; its mirror row is in the window
        const int yg = a.y_first + e;
        if (yg == 0 && !a.spole) c &= ~0x02020202u;
        if (yg == a.H - 1 && !a.npole) c &= ~0x01010101u;
        // ... and the same bit seen from the mirror row (after the swap: row -1's north, row H's south neighbour is the
        // edge row, whose values are the mirror row's own).  The counts below need a set bit to mean "that neighbour
        // holds +0.0"; the bit-field selects of the other kernels did not care (pK | pC with pK == pC)
        if (yg == -1 && !a.spole) c &= ~0x01010101u;
        if (yg == a.H && !a.npole) c &= ~0x02020202u;
        const unsigned long long bm = __builtin_amdgcn_ballot_w64((c & 0x1F1F1F1Fu) != 0), bl = __builtin_amdgcn_ballot_w64((c & 0x10101010u) != 0x10101010u);
        mix = (unsigned)((bm | (0ull - bm)) >> 63);          // any lane with a code bit
        land = (unsigned)((bl | (0ull - bl)) >> 63) ^ 1u;    // no lane with a water cell
        mix &= ~land;
        if (mix) {
            // code byte of a cell: bit 0 N, 1 S, 2 E, 3 W neighbour is land, bit 4 the cell itself.  v_perm_b32 as a byte
            // table: {E,W} -> fp4 code of their count; {N,S} likewise, and -1.0 for a land cell (E2M1: 2 = 1.0, 4 = 2.0, 0xA = -1.0)
            const unsigned c2 = c >> 2;
            const unsigned we = __builtin_amdgcn_perm(0u, 0x04020200u, c2 & 0x03030303u);
            const unsigned sn = __builtin_amdgcn_perm(0x0A0A0A0Au, 0x04020200u, (c & 0x03030303u) | (c2 & 0x04040404u));
            // nibbles into pair order: bytes {we c0, we c1, sn c0, sn c1} | {we c3, we c2, sn c3, sn c2} << 4
            cring[slot][lane] = __builtin_amdgcn_perm(sn, we, 0x05040100u) | (__builtin_amdgcn_perm(sn, we, 0x06070203u) << 4);
            // the right-hand side of a land cell becomes -0.0: x = 0 * t - (-0.0) is then +0.0 whatever the sign of t
            v4f bq = z.bs[K][j];
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.x) : "v"(code_mask<4>(c, 0)), "v"(0x80000000u), "v"(bq.x));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.y) : "v"(code_mask<12>(c, 0)), "v"(0x80000000u), "v"(bq.y));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.z) : "v"(code_mask<20>(c, 0)), "v"(0x80000000u), "v"(bq.z));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.w) : "v"(code_mask<28>(c, 0)), "v"(0x80000000u), "v"(bq.w));
            st_park_pairs(&bring[slot][lane], bq);
            return;
        }
    }
    // (an open row among coast rows goes through the coast formula with the rest of its step: counts of zero)
    cring[slot][lane] = 0u;                                   // (a land row's entry is never read: no test)
    st_park_pairs(&bring[slot][lane], z.bs[K][j]);
}

// class of the row in staging slot [K][j] (persistent form): rows beyond the grid or the stored band count as mixed
template <int K>
__device__ __forceinline__ unsigned st_class_p(const st_args &a, const st_state &z, int j, int e)
{
    const int yg = a.y_first + e;
    if (!ST_DBG(a, 64)) return 2u;                            // the codes decide (st_arrive)
    const bool inside = yg >= 0 && yg < a.H && e >= 0 && e < a.L;
    // (volatile: left to the scheduler, this read -- and with it the wait for the staging set's loads -- moves a whole step
    // up, to where the set before it has only just been taken)
    unsigned c;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(c) : "v"(z.ks[K][j]) : "memory");
    return inside ? c : 2u;
}

// ---- twin units (round 4): two units of a strip that START on the same seam and stream away from it ------------------------
// A unit of n rows streams n + 16: eight rows beyond each end, on which its levels shrink back to the rows it writes -- a
// 28-row unit (the band of an 8-way split has ~28 rows per wave slot) runs 22 steps for 14 steps' worth of rows.  The halo
// exists because the unit's neighbour computes ITS rows somewhere else at some other time.  Two waves of one workgroup that
// start on the two sides of a seam row at the same moment need no halo between them: the northward wave's level s needs, of
// the other side, only level s-1 of the ONE row below the seam, which is the southward wave's first row of that level (and
// the other way round), made in the same or the previous step.  So the two hand each other their first row of every level
// through LDS while their pipelines fill -- the pipeline of such a unit is full after FOUR steps instead of eight
// (level s has its first row at step s/2: there is no shrinking edge to wait for), it streams n + 8 rows, and after step 3 the
// two waves never meet again.  At step t (0..3): level 2t makes the unit's rows (0, 1); its row 0 goes to the partner, a
// barrier, and the partner's is this wave's row -1 of level 2t, which level 2t + 1 needs for row 0 in this same step; that row
// 0 goes over likewise, and the partner's becomes row -1 of level 2t + 1 for level 2t + 2 in the next step.  Nine barriers
// per unit, all in its first five steps (the ninth: k_jacobi_stream_twin, the ring slots the rows travel through).  The southward wave is the same program on the mirrored view (above).
// Same operations on the same values as a unit with a halo: bit-identical (tests/stream_worker.py `twin`).
template <int S, int K>
__device__ __forceinline__ void st_twin_hand_over(st_state &z, v4f (*xme)[64], const v4f (*xpeer)[64], int lane, bool second)
{
    // (LDS only: the wait leaves the wave's global loads in flight, which __syncthreads()'s fence need not)
    if (!second) {
        xme[0][lane] = z.w[S][K][0];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        z.w[S][K ^ 1][1] = xpeer[0][lane];
    } else {
        xme[1][lane] = z.w[S][K][1];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        z.w[S][K][0] = xpeer[1][lane];
    }
}
template <int S, int K, int DV>
__device__ __forceinline__ void st_twin_level(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const v4f *ktab,
                                              int lane, int t, int e_t, int r0, int r1, bool lane_ok, int xcol, v4f (&bq)[9], v4f (&kq)[9], int kb,
                                              v4f (*xme)[64], const v4f (*xpeer)[64])
{
    if (S <= 2 * t + 1) {
        st_level<S, K, false, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - S, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (S < 8) {
            if (S == 2 * t) st_twin_hand_over<S & 7, K>(z, xme, xpeer, lane, false);
            else if (S == 2 * t + 1) st_twin_hand_over<S & 7, K>(z, xme, xpeer, lane, true);
        }
    }
}

template <int K, bool P = false, int DV = 0, int TW = 0>
__device__ __forceinline__ void st_step(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const v4f *ktab, int lane, int t,
                                        int e0, int nsteps, int r1, unsigned cls_lo, unsigned cls_hi, int col, int colp, bool lane_ok, int xcol, int tx = 0,
                                        v4f (*xme)[64] = nullptr, const v4f (*xpeer)[64] = nullptr)
{
    const int e_t = e0 + 2 * t;                              // first row of the pair that has just arrived
    const int slotB = z.slot + 1;                            // (ST_RING is even: a pair never wraps)
    unsigned landA, mixA, landB, mixB;
    const unsigned clsA = P ? st_class_p<K>(a, z, 0, e_t) : z.fc[K][0], clsB = P ? st_class_p<K>(a, z, 1, e_t + 1) : z.fc[K][1];
#ifdef ST_EXPERIMENTS
    // DMG_ST_STEPPROF: slots 22..42 = the clock when the even step's rows have arrived (its wait for the staged loads is over),
    // slots 43..62 = when the even step's levels are done, of a unit's first 20 iterations (slots 0..21: top of the iteration)
    if (!P && K == 0 && a.stepprof && lane == 0 && t < 40) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // (the even step's six loads are the oldest of the twelve in flight)
        a.stepprof[(size_t)(TW ? 2 * blockIdx.x + (threadIdx.x >> 6) : blockIdx.x) * 64 + 22 + t / 2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // (twin units: the partner has read the last row handed over -- ring slots 10 and 11 are the unit's own from here on)
    if (TW && K == 0 && t == TB_T / 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    st_arrive<K>(a, z, bring, cring, lane, 0, e_t, clsA, z.slot, landA, mixA);
    st_arrive<K>(a, z, bring, cring, lane, 1, e_t + 1, clsB, slotB, landB, mixB);
    if (TW && K == 0 && t == 0) st_twin_hand_over<0, K>(z, xme, xpeer, lane, false);      // "level 0": the partner's first row is this unit's row -1
    z.landbits = (z.landbits << 2) | (landA << 1) | landB;
    z.mixbits = (z.mixbits << 2) | (mixA << 1) | mixB;
    if (DV) z.badbits = (z.badbits << 2) | ((clsA >> 1) & 2u) | ((clsB >> 2) & 1u);
    const unsigned special = DV ? z.badbits : 0u;           // rows that need the tested path like an all-land row does
    // the staging set is free again: the pair of step t + 2 (past the window's end the rows are clamped and unused)
    if (!P) { z.fc[K][0] = st_class4(cls_lo, 2 * t + 4); z.fc[K][1] = st_class4(cls_lo, 2 * t + 5); }
    st_fetch<P>(a, z, K, 0, e_t + 4, col, colp, tx, P ? 2u : z.fc[K][0]);
    st_fetch<P>(a, z, K, 1, e_t + 5, col, colp, tx, P ? 2u : z.fc[K][1]);
    // level 1: rows e_t - 1 (in the ring since the last step) and e_t (just written)
    // (a twin unit's level 1 computes window row 0 in step 0: its table index is not clamped -- row -1's entry is the 16 bytes in
    // front of the table, read and not used)
    const int s1 = st_wrap(z.slot - 1), r1w = TW ? 2 * t - 1 : max(2 * t - 1, 0);
    v4f bq[9], kq[9];
    bq[1] = bring[s1][lane]; bq[0] = bring[z.slot][lane]; kq[1] = ktab[st_kt<P>(r1w)]; kq[0] = ktab[st_kt<P>(r1w + 1)];
    int kb = 0;
    if (!P) asm volatile("v_mov_b32 %0, %1" : "=v"(kb) : "s"((2 * t - 8) * 16));
    // no land and no coast in any of the ten rows in flight (every step of a strip in open water): the eight levels without
    // a test or a branch between them -- one block for the scheduler; a branch per level visit costs the kernel ~3 %
    // A unit's first eight steps fill the pipeline: level S has its first rows to compute at step S.  Those steps take the
    // tested path whatever their rows are, and there a level that has nothing to do yet is skipped outright (round 3: per-step
    // clocks showed a skipped level visit costing 0.25 us through the all-land path -- as much as a real one -- and a unit's
    // eight fill steps 2.1 us each; with the skip a unit is ~7 us shorter)
    const bool filled = P || t >= (TW ? TB_T / 2 : TB_T);
    const int r0 = TW ? e0 : e0 + TB_T;                      // the unit's first output row
    if (filled && !((z.landbits | z.mixbits | special) & 0x3FFu)) {
        st_level<1, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<2, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<3, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<4, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<5, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<6, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<7, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<8, K, P, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    } else if (filled && !((z.landbits | special) & 0x3FFu)) {
        // coast among the ten rows, but no row that is land everywhere (coast rows come in long runs: 18 % of the survey
        // terrain's steps): every row through the coast formula, open rows with counts of zero -- again no test, no branch
        st_level<1, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<2, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<3, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<4, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<5, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<6, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<7, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<8, K, P, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    } else if (TW) {
        st_twin_level<1, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<2, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<3, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<4, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<5, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<6, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<7, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
        st_twin_level<8, K, DV>(a, z, bring, cring, ktab, lane, t, e_t, r0, r1, lane_ok, xcol, bq, kq, kb, xme, xpeer);
    } else {
        if (P || 1 <= t) st_level<1, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 2 <= t) st_level<2, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 3 <= t) st_level<3, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 4 <= t) st_level<4, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 5 <= t) st_level<5, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 6 <= t) st_level<6, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 7 <= t) st_level<7, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (P || 8 <= t) st_level<8, K, P, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    }
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
#ifdef ST_EXPERIMENTS
    if (!P && K == 0 && a.stepprof && lane == 0 && t < 40) a.stepprof[(size_t)(TW ? 2 * blockIdx.x + (threadIdx.x >> 6) : blockIdx.x) * 64 + 43 + t / 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- open-water units (round 4; measured, NOT the default: DMG_ST_OPEN_UNITS=1 switches them on) -----------------------
// A step's eight levels are 386 instructions, 350 of them the formula; around them a generic step spends another ~180 -- row
// classes looked up and tested, three row pointers advanced and an open row's code load redirected, land / coast / divisor
// histories built and the mode chosen, stores range-tested: mostly scalar.  Round 3 read the kernel as bound by instruction
// issue of every kind and named exactly these instructions as the next lever.  The host knows what the rows are: a unit whose
// whole window is plain open water with exact divisor pairs, inside the grid and the stored band (75-80 % of the survey
// terrain's live rows at any unit height, every interior unit of an all-ocean grid) is flagged in the unit list and runs
// through this program instead: a row is two loads through one scalar base per array and a lane offset built from a scalar row
// offset (its codes are never looked at), its arrival two packed moves and two LDS writes, level 8 stores without a range
// test, nothing is classified or tested, no count ring, no histories: ~440 instead of ~590 instructions per step.  A region of
// its own -- no step ever changes form -- because the same path INSIDE the generic loop (as a fourth alternative per step, or
// as a six-step block with literal ring slots) pushed the register allocation of a 251-VGPR kernel over the edge: 113-257
// spilled registers, and a scratch reload waits with vmcnt(0), which drains the two-steps-ahead loads: 3.6 -> 1.7 M.
// The same level code on the same values: bit-identical (tests/stream_worker.py `open`, eight variants).
// What it measured (profiles/r04_open_units.txt, same-box A/B, all-ocean 16384x8192): scalar instructions per launch 40.4 M ->
// 13.6 M, vector 119.2 M -> 113.4 M, the waves' summed cycles -16 % -- and the LAUNCH not a percent shorter (368.9 vs 364.9 us;
// survey terrain 280.0 vs 271.8 us).  The kernel is not issue-bound: its launches move 1.36 (survey) / ~1.9 GB (all-ocean) at
// 5.0-5.1 TB/s where whole 1-KB rows fetched once reach 5.5-5.6 TB/s (MI355X_MICROARCH.md), a 16384x1024 band 0.33 GB in
// 65 us likewise; waves that finish their arithmetic sooner wait for the same bytes.  What is left to gain is bytes
// (DESIGN.md 8), not instructions; the program stays in the file, tested, for the day the bytes come down.
template <int K, int DV, int PHASE>      // PHASE 0: the pipeline fills (level S starts at step S); 1: steady; 2: the unit's last steps (level 8 range-tested)
__device__ __forceinline__ void st_open_step(const st_args &a, st_state &z, v4f (*bring)[64], const v4f *ktab, int lane, int t, int e_t, int r0, int r1,
                                             bool lane_ok, int xcol, unsigned col4, unsigned &rowb, const char *pb, const char *bb, const char *sb)
{
    // arrival: pressure into level 0 in pair order (opaque moves: st_arrive says why), the right-hand side into its ring slots
#pragma unroll
    for (int j = 0; j < 2; j++) {
        v2f lo, hi;
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(lo) : "v"(z.ps[K][j].xy), "v"(z.ps[K][j].zw) : "memory");
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(hi) : "v"(z.ps[K][j].xy), "v"(z.ps[K][j].zw) : "memory");
        z.w[0][K][j] = v4f{ lo.x, lo.y, hi.x, hi.y };
        st_park_pairs(&bring[z.slot + j][lane], z.bs[K][j]);
    }
    // the pair of step t + 2
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
    constexpr int M = PHASE == 1 ? 3 : 1;
    if (PHASE != 0 || 1 <= t) st_level<1, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 2 <= t) st_level<2, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 3 <= t) st_level<3, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 4 <= t) st_level<4, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 5 <= t) st_level<5, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 6 <= t) st_level<6, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 7 <= t) st_level<7, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
    // (voA, voB address rows e_t + 4, e_t + 5; level 8's rows are e_t - 8, e_t - 7: sb = pout - 12 rows)
    if (PHASE != 0) st_level<8, K, false, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb, sb, voA, voB);
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
}

template <int DV>
__device__ __forceinline__ void st_open_unit(const st_args &a, v4f (*bring)[64], v4f *ktab, int lane, int tx, int r0, int r1)
{
    const int W = a.W;
    const int nsteps = (((r1 - r0) + 2 * TB_T + 3) / 4) * 2;
    const int e0 = r0 - TB_T;
    int col = tx * TB_OW - TB_T + 4 * lane;
    const int xcol = col;
    col %= W; if (col < 0) col += W;
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
    const unsigned col4 = (unsigned)col * 4u;
    const char *pb = reinterpret_cast<const char *>(a.pin), *bb = reinterpret_cast<const char *>(a.b);
    const char *sb = reinterpret_cast<const char *>(a.pout) - (size_t)12 * W * 4;
    unsigned rowb = (unsigned)e0 * (unsigned)W * 4u;          // (the host flags a unit open only if its window and the rows requested past it lie inside the band)
    st_state z;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const unsigned voA = rowb + col4, voB = voA + (unsigned)W * 4u;
        z.ps[k][0] = *reinterpret_cast<const v4f *>(pb + voA); z.bs[k][0] = *reinterpret_cast<const v4f *>(bb + voA);
        z.ps[k][1] = *reinterpret_cast<const v4f *>(pb + voB); z.bs[k][1] = *reinterpret_cast<const v4f *>(bb + voB);
        rowb += (unsigned)W * 8u;
    }
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0; z.mixbits = 0; z.badbits = 0; z.slot = 0; z.plain = true;
    {
        const int n = 2 * nsteps + 2;
        v4f kt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) kt[q] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)(e0 + min(lane + 64 * q, n - 1)) * 8));
#pragma unroll
        for (int q = 0; q < 4; q++) if (lane + 64 * q < n) ktab[lane + 64 * q] = kt[q];
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    int t = 0;
    for (; t < TB_T; t += 2) {
        st_open_step<0, DV, 0>(a, z, bring, ktab, lane, t, e0 + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
        st_open_step<1, DV, 0>(a, z, bring, ktab, lane, t + 1, e0 + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
    }
    // steady: both steps' level-8 rows are rows the unit writes: e0 + 2 (t + 1) - 8 + 1 < r1.  The first pair is peeled off the
    // loop: a step waits for the loads of two steps ago with a counted vmcnt, and the count the compiler can prove at the loop's
    // head is the smaller of its two predecessors' -- coming from the fill steps, which store nothing, it would be vmcnt(4)
    // (the loads of ONE step ago must be back too) for every iteration instead of vmcnt(8)
    if (t + 2 <= nsteps && e0 + 2 * t - 5 < r1) {
        st_open_step<0, DV, 1>(a, z, bring, ktab, lane, t, e0 + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
        st_open_step<1, DV, 1>(a, z, bring, ktab, lane, t + 1, e0 + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
        t += 2;
        asm volatile("" ::: "memory");
    }
    for (; t + 2 <= nsteps && e0 + 2 * t - 5 < r1; t += 2) {
        st_open_step<0, DV, 1>(a, z, bring, ktab, lane, t, e0 + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
        st_open_step<1, DV, 1>(a, z, bring, ktab, lane, t + 1, e0 + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
    }
    for (; t < nsteps; t += 2) {
        st_open_step<0, DV, 2>(a, z, bring, ktab, lane, t, e0 + 2 * t, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
        st_open_step<1, DV, 2>(a, z, bring, ktab, lane, t + 1, e0 + 2 * t + 2, r0, r1, lane_ok, xcol, col4, rowb, pb, bb, sb);
    }
}

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
#define ST_TWIN_MAXROWS 232
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
#define STP_BIAS 1024             // rows are stored + STP_BIAS (a window starts 8 rows below its segment)
#define STP_MIN_STEAL 24          // rows left (not yet taken in by the owner) below which a segment is left alone
// A segment's word: strip (12 bits) | first row + STP_BIAS (26 bits) | end row + STP_BIAS (26 bits).  Everything a thief
// needs to stay inside the segment is in this one word; the owner's progress (seg_cur) is advisory -- whatever a thief
// reads there is clamped into [first - 8, end], so a stale or not yet written value only makes the split less even.
#define STP_FIELD 26
#define STP_MASK ((1u << STP_FIELD) - 1u)
__host__ __device__ __forceinline__ unsigned long long stp_word(int tx, int lo, int hi)
{
    return ((unsigned long long)(unsigned)tx << (2 * STP_FIELD)) | ((unsigned long long)(unsigned)(lo + STP_BIAS) << STP_FIELD) | (unsigned)(hi + STP_BIAS);
}
__device__ __forceinline__ int stp_hi(unsigned long long w) { return (int)((unsigned)w & STP_MASK) - STP_BIAS; }
__device__ __forceinline__ int stp_lo(unsigned long long w) { return (int)((unsigned)(w >> STP_FIELD) & STP_MASK) - STP_BIAS; }
__device__ __forceinline__ int stp_tx(unsigned long long w) { return (int)(w >> (2 * STP_FIELD)); }

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
#define STN_EW 128
#define STN_OW (STN_EW - 2 * TB_T)
#define ST_PAD 16                 // rows of padding on both ends of the coefficient table: window rows -16 .. L+15 need no mapping

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
bool ocean_stream_enabled()
{
    static const bool on = [] { const char *e = getenv("DMG_JACOBI_STREAM"); return !e || atoi(e) != 0; }();
    return on;
}

static int st_unit_rows()                                 // DMG_ST_ROWS=<n>: fixed unit height (experiments); default 0 = chosen per launch geometry
{
    static const int rows = [] {
        const char *e = getenv("DMG_ST_ROWS");
        int v = e ? atoi(e) : 0;
        return v >= 1 && v <= ST_MAXROWS ? v : 0;
    }();
    return rows;
}

namespace {
struct st_unit { int tx, r0, r1; float cost; int flip = 0;         // flip: the unit streams southwards (k_jacobi_stream's second view)
                 int rlo = -1;                                     // >= 0: a pair of twin units (k_jacobi_stream_twin) -- rows [r0, r1) northwards and [rlo, r0) southwards from the seam r0
                 float cost_n = 0.0f, cost_s = 0.0f;               // twin pair: the northward and the southward unit's cost (cost = 16 x the dearer)
                 int south_first = 0;                              // twin pair: the southward unit is the workgroup's first wave (st_place_twins)
                 int dummy = 0;                                    // a workgroup that ends at once (keeps its successors on the CUs st_place_twins meant them for)
                 int first() const { return rlo >= 0 ? rlo : r0; } };
}

// What the steps of a unit are made of, as the kernel will run them (st_step): a step whose ten rows in flight are all open
// water runs its eight levels as one block without a test; one with coast rows but no row that is land everywhere runs every
// row through the coast formula; every other step -- the pipeline's fill, a land row in flight -- takes the tested path, level by
// level and row by row.  cls: the strip's row classes (null: all mixed), rows outside [0, L) and outside the grid count as mixed
// as in the kernel; flip: the unit runs on the mirrored view (rows r0, r1 are that view's); twin: its window starts on r0.
struct st_features { int fast = 0, coast = 0, tested = 0; int t_open = 0, t_mixed = 0, t_land = 0;     // tested: steps past the fill; t_*: their level-rows by class
                     int f_open = 0, f_mixed = 0, f_land = 0; };                                    // the fill steps' level-rows
static st_features st_unit_features(const unsigned char *cls, int L, int y_first, int H, int r0, int r1, bool flip, bool twin)
{
    st_features f;
    const int e0 = twin ? r0 : r0 - TB_T, nsteps = (((r1 - r0) + (twin ? TB_T : 2 * TB_T) + 3) / 4) * 2;
    const int yf = flip ? H - y_first - L : y_first;
    auto cl = [&](int lu) -> int {
        if (lu < e0) return 1;                              // rows before the window: "land" (st_state::landbits)
        const int yg = yf + lu;
        if (!cls || lu < 0 || lu >= L || yg < 0 || yg >= H) return 2;
        return cls[flip ? L - 1 - lu : lu];
    };
    for (int t = 0; t < nsteps; t++) {
        const int newest = e0 + 2 * t + 1;
        bool land = false, mix = false;
        for (int k = 0; k < 10; k++) { const int c = cl(newest - k); land |= c == 1; mix |= c == 2; }
        const bool filled = t >= (twin ? TB_T / 2 : TB_T);
        if (filled && !land && !mix) f.fast++;
        else if (filled && !land) f.coast++;
        else {
            if (filled) f.tested++;
            for (int S = 1; S <= 8; S++) {
                if (twin ? S > 2 * t + 1 : S > t) continue;
                for (int k = S - 1; k <= S; k++) {
                    const int c = cl(newest - 1 - k);
                    if (filled) (c == 0 ? f.t_open : (c == 2 ? f.t_mixed : f.t_land))++;
                    else (c == 0 ? f.f_open : (c == 2 ? f.f_mixed : f.f_land))++;
                }
            }
        }
    }
    return f;
}

// A unit's predicted time in "open-water steps", from what its steps are made of.  The weights are the marginal costs that
// per-unit timings of a 16384 x 1024 band give when regressed on these counts (DMG_ST_PROF, six launches of two cuts, younger
// and older wave of a SIMD apart; profiles/r04_twin_units.txt): a step through the coast formula 1.6 .. 1.7 open-water steps;
// a tested step (a land row in flight) per row and level it visits 0.05 for an open row, 0.14 for a coast row and 0.11 for a
// LAND row -- a land row computes nothing, but a level visit on the tested path costs its tests and branches whatever its
// rows are, which is what round 3's step clocks showed for skipped levels; the fill steps 0.37 + the same per row and level
// (step clocks: 1.0 + 0.12 us per level-row against 2.7 us for a full step).  r0, r1 in the band's own numbering; flip: the
// unit streams southwards; twin: its window starts on its first row (no halo there).
struct st_weights {
    float coast, t_open, t_mixed, t_land;
    static const st_weights &get()
    {
        static const st_weights w = [] {
            auto env = [](const char *name, float dflt) { const char *e = getenv(name); return e ? (float)atof(e) : dflt; };
            return st_weights{ env("DMG_ST_WCOAST", 1.7f), env("DMG_ST_WTOPEN", 0.05f), env("DMG_ST_WTMIXED", 0.14f), env("DMG_ST_WTLAND", 0.11f) };
        }();
        return w;
    }
};
static float st_exact_cost(const dmg_ocean *o, int tx, int r0, int r1, bool flip, bool twin)
{
    const st_weights &w = st_weights::get();
    const unsigned char *cls = o->rowclass_host.empty() ? nullptr : o->rowclass_host.data() + (size_t)tx * o->L;
    const st_features f = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, flip ? o->L - r1 : r0, flip ? o->L - r0 : r1, flip, twin);
    const float fill_steps = twin ? 4.0f : 8.0f;
    return 0.7f + (float)f.fast + w.coast * (float)f.coast + 0.37f * fill_steps + w.t_open * (float)(f.t_open + f.f_open) + w.t_mixed * (float)(f.t_mixed + f.f_mixed) +
           w.t_land * (float)(f.t_land + f.f_land);
}

// what the dispatch simulation charges a unit of rows [r0, r1) of a strip (cls: the strip's row classes, or null): its
// level-rows, a coast row counted 1.4 times an open one and a land row a tenth (what the kernel spends on them), plus what a
// unit costs before and after its rows
static float st_unit_cost(const unsigned char *cls, int L, int r0, int r1, int variant, int twin = 0)    // twin 1: no halo below r0 (a unit that starts on a seam), 2: none above r1
{
    static const float w_mixed = [] { const char *e = getenv("DMG_ST_WMIXED"); return e ? (float)atof(e) : 1.4f; }();    // experiments: the cost model's weights (1.4: best of a sweep 1.0 .. 1.6 on the survey terrain; the per-unit timings fit 1.23 .. 1.40)
    static const float w_land = [] { const char *e = getenv("DMG_ST_WLAND"); return e ? (float)atof(e) : 0.1f; }();
    float cost = 0.0f;
    for (int r = twin == 1 ? r0 : r0 - TB_T; r < (twin == 2 ? r1 : r1 + TB_T); r++) {
        const int c = (cls && r >= 0 && r < L) ? cls[r] : 2;
        const int below = twin == 1 ? TB_T : r - (r0 - TB_T), above = twin == 2 ? TB_T : (r1 + TB_T - 1) - r;
        const int depth = std::min(std::min(below, above), TB_T);   // levels that still need this row
        cost += (c == 1 ? w_land : (c == 2 ? w_mixed : 1.0f)) * (float)depth;
    }
    // + what a unit costs before and after its rows: dispatch, the class and coefficient tables, the first loads' round
    // trip, the last stores.  Per-unit timings (DMG_ST_PROF) fit duration = 0.152 us x cost + 22 us: ~140 cost units,
    // not the 4 the first version assumed -- which made the simulation prefer many short units
    static const float unit_cost = [] { const char *e = getenv("DMG_ST_UNITCOST"); return e ? (float)atof(e) : 140.0f; }();
    // an open-water unit (st_open_unit: no row of its window is anything but open water) spends ~440 instead of ~590
    // instructions per step
    static const float w_openunit = [] { const char *e = getenv("DMG_ST_WOPEN"); return e ? (float)atof(e) : 1.0f; }();     // (experiments; 0.78 is what their instruction count suggests and what made the cut worse)
    if (cls && !variant && w_openunit != 1.0f) {
        bool all_open = r0 - TB_T >= 0 && r1 + TB_T + 6 <= L;
        for (int r = r0 - TB_T; all_open && r < r1 + TB_T + 6; r++) all_open = cls[r] == 0;
        if (all_open) cost *= w_openunit;
    }
    return cost + unit_cost;
}

// Units of a launch over local rows [lo, hi) for one unit height: every strip cut into segments of `seg` rows, each trimmed
// to its first / last row that is not land everywhere; empty ones are dropped.  cost ~ the unit's level-rows, a coast row
// counted 1.4 times an open one and a land row a tenth (what the kernel spends on them).
static void st_cut(const dmg_ocean *o, int variant, int lo, int hi, int seg, int part, int zone_lo, int zone_hi, std::vector<st_unit> &units,
                   int seg2 = 0, int split = 0, int edge_seg = 0)
{
    const int ow = variant ? STN_OW : TB_OW;
    const std::vector<unsigned char> &table = variant ? o->rowclass_n_host : o->rowclass_host;
    const int tiles_x = (o->grid.W + ow - 1) / ow, L = o->L;
    const bool classes = !table.empty();
    units.clear();
    // rows [lo, split) in segments of `seg` rows, the rest in (shorter) segments of `seg2` rows: the short units fill
    // the last round of the dispatch
    if (!seg2) split = hi;
    // edge_seg > 0 (the halves of a split launch): the send zones [lo, zone_lo) and [zone_hi, hi) are cut into units of edge_seg
    // rows of their own (4: measured 4 .. 32) and the rows between them as the whole launch cuts them, anchored at zone_lo.  An edge unit of 8 rows
    // is done in ~25 us, one of 28 in ~45: the exchange behind the edge units starts that much earlier, and what it can hide
    // behind is the interior units of one launch (~57 us on a 1024-row band) -- see profiles/r03_fake_link.log
    const bool zones = edge_seg > 0 && zone_lo >= lo && zone_hi <= hi && zone_lo <= zone_hi;
    const int mid_lo = zones ? zone_lo : lo, mid_hi = zones ? zone_hi : hi;
    // every other segment streams southwards (k_jacobi_stream): neighbouring units then start, or end, on the rows they share
    static const int flip_env = [] { const char *e = getenv("DMG_ST_FLIP"); return e ? atoi(e) : -1; }();
    const bool flipping = variant == 0 && (flip_env >= 0 ? flip_env != 0 : ST_FLIP_DEFAULT != 0);
    int segi = 0;
    for (int s0 = lo; s0 < hi; segi++) {
        int s1;
        if (s0 < mid_lo) s1 = std::min(s0 + edge_seg, mid_lo);
        else if (s0 >= mid_hi) s1 = std::min(s0 + edge_seg, hi);
        else {
            const int h = s0 < split ? seg : seg2;
            s1 = std::min(std::min(s0 + h, s0 < split ? split : hi), mid_hi);
            if (s1 <= s0) s1 = std::min(s0 + seg, mid_hi);                  // (a split that lies below the middle zone)
        }
        const int s0_this = s0;
        s0 = s1;
        {
            const int s0 = s0_this;
        for (int tx = 0; tx < tiles_x; tx++) {
            int r0 = s0, r1 = s1;
            const unsigned char *cls = classes ? table.data() + (size_t)tx * L : nullptr;
            if (cls) {
                while (r0 < r1 && cls[r0] == 1) r0++;
                while (r1 > r0 && cls[r1 - 1] == 1) r1--;
            }
            if (r0 >= r1) continue;
            if (part) {
                const bool edge = r0 < zone_lo || r1 > zone_hi;
                if (edge != (part == 1)) continue;
            }
            const float cost = st_unit_cost(cls, L, r0, r1, variant), unit_cost = 0.0f;
            units.push_back(st_unit{ tx, r0, r1, cost + unit_cost, (flipping && (segi & 1)) ? 1 : 0 });
        }
        }
    }
}

// A cut for launches of about one round (round 4).  A band of a thousand rows has about as many units as the chip has wave
// slots, so the launch lasts as long as its SLOWEST unit -- and with one height for every strip the slowest are the coast
// units (a coast row costs 1.4 of an open one): of a 16384 x 1024 band's 1831 units the median takes 59 us, the launch 65
// (DMG_ST_PROF), while 11 % of the slots stay empty.  Here the middle zone of the launch is cut strip by strip into `want`
// units of about EQUAL COST: every stretch of water (land runs of 24 rows and more are left out) starts as one unit, and the
// stretch whose units are the most expensive gets one more until `want` are dealt out; a stretch's units then split its rows
// at equal cumulative cost.  The send zones of a split launch keep their short units (st_cut).  Which cut a launch takes
// is still decided by the simulated dispatch (st_units): this one is a candidate at one, two and three rounds' worth of units.
// twin: the stretches are cut into PAIRS of units that start on a common seam (k_jacobi_stream_twin; `want` still counts units)
static void st_cut_balanced(const dmg_ocean *o, int lo, int hi, int want, int part, int zone_lo, int zone_hi, std::vector<st_unit> &units, int edge_seg = 0,
                            bool twin = false)
{
    const std::vector<unsigned char> &table = o->rowclass_host;
    const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW, L = o->L;
    const bool classes = !table.empty();
    static const int flip_env = [] { const char *e = getenv("DMG_ST_FLIP"); return e ? atoi(e) : -1; }();
    const bool flipping = flip_env >= 0 ? flip_env != 0 : ST_FLIP_DEFAULT != 0;
    const bool zones = edge_seg > 0 && zone_lo >= lo && zone_hi <= hi && zone_lo <= zone_hi;
    const int mid_lo = zones ? zone_lo : lo, mid_hi = zones ? zone_hi : hi;
    units.clear();
    int n_all = 0;                                           // units of the whole launch so far (a part keeps only its own, but both parts cut alike)
    auto emit = [&](int tx, const unsigned char *cls, int r0, int r1, int flip) {
        if (cls) {
            while (r0 < r1 && cls[r0] == 1) r0++;
            while (r1 > r0 && cls[r1 - 1] == 1) r1--;
        }
        if (r0 >= r1) return;
        n_all++;
        if (part) {
            const bool edge = r0 < zone_lo || r1 > zone_hi;
            if (edge != (part == 1)) return;
        }
        units.push_back(st_unit{ tx, r0, r1, st_unit_cost(cls, L, r0, r1, 0), flip });
    };
    // a pair of twin units around the seam: rows [rlo, seam) southwards, [seam, r1) northwards; only their far ends are trimmed
    auto emit_twin = [&](int tx, const unsigned char *cls, int rlo, int seam, int r1) {
        if (cls) {
            while (rlo < seam && cls[rlo] == 1) rlo++;
            while (r1 > seam && cls[r1 - 1] == 1) r1--;
        }
        if (rlo >= r1) return;
        n_all += 2;
        if (part) {
            const bool edge = rlo < zone_lo || r1 > zone_hi;
            if (edge != (part == 1)) return;
        }
        const float cn = st_exact_cost(o, tx, seam, r1, false, true), cs = st_exact_cost(o, tx, rlo, seam, true, true);
        st_unit u{ tx, seam, r1, 16.0f * std::max(cn, cs), 0 };
        u.rlo = rlo; u.cost_n = cn; u.cost_s = cs;
        units.push_back(u);
    };
    // the send zones: short units of their own, as in st_cut -- in a twin cut as pairs of such units around a seam (a unit of four
    // rows then runs six steps instead of ten: the edge half of a split launch is what the exchange waits for).  DMG_ST_TWIN_EDGES=0:
    // plain short units in a twin cut's zones as well
    const bool twin_edges = [] { const char *e = getenv("DMG_ST_TWIN_EDGES"); return !e || atoi(e) != 0; }();
    if (zones)
        for (int tx = 0; tx < tiles_x; tx++) {
            const unsigned char *cls = classes ? table.data() + (size_t)tx * L : nullptr;
            if (twin && twin_edges) {
                auto zone_pairs = [&](int z0, int z1) {
                    for (int s0 = z0; s0 < z1; s0 += 2 * edge_seg) {
                        const int s2 = std::min(s0 + 2 * edge_seg, z1), seam = std::min(s0 + edge_seg, s2);
                        emit_twin(tx, cls, s0, seam, s2);
                    }
                };
                zone_pairs(lo, mid_lo);
                zone_pairs(mid_hi, hi);
                continue;
            }
            int k = 0;
            for (int s0 = lo; s0 < mid_lo; s0 += edge_seg, k++) emit(tx, cls, s0, std::min(s0 + edge_seg, mid_lo), flipping && (k & 1));
            for (int s0 = mid_hi; s0 < hi; s0 += edge_seg, k++) emit(tx, cls, s0, std::min(s0 + edge_seg, hi), flipping && (k & 1));
        }
    // stretches of water in the middle zone
    struct stretch { int tx, r0, r1, k; double cost; };
    std::vector<stretch> st;
    static const double w_mixed = [] { const char *e = getenv("DMG_ST_WMIXED"); return e ? atof(e) : 1.4; }();
    // twin units: what a row adds to its unit is what the STEP it arrives in costs (st_exact_cost's weights: its ten rows in flight
    // decide the step's form), so that a stretch's weight, the number of units it is given and its first cut are priced the way
    // the finished units will be.  (Priced row by row -- a coast row 1.4, a land row 0.1 -- the stretches along the coasts got
    // too few units: forcing the coast weight to 2.4 brought the 8-way band from 52.9 to 48.1 us, profiles/r04_twin_units.txt.)
    std::vector<float> step_w;
    if (twin && classes) {
        const st_weights &w = st_weights::get();
        step_w.assign((size_t)tiles_x * L, 1.0f);
        for (int tx = 0; tx < tiles_x; tx++) {
            const unsigned char *cls = table.data() + (size_t)tx * L;
            auto cl = [&](int r) { return r >= 0 && r < L ? (int)cls[r] : 2; };
            for (int r = std::max(lo - 1, 0); r < std::min(hi + 1, L); r++) {
                bool land = false, mix = false;
                for (int k = 0; k < 10; k++) { const int c = cl(r - k); land |= c == 1; mix |= c == 2; }
                float v;
                if (land) {
                    v = 0.0f;
                    for (int S = 1; S <= 8; S++)
                        for (int k = S; k <= S + 1; k++) { const int c = cl(r - k); v += c == 0 ? w.t_open : (c == 2 ? w.t_mixed : w.t_land); }
                } else v = mix ? w.coast : 1.0f;
                step_w[(size_t)tx * L + r] = v;
            }
        }
    }
    auto roww = [&](const unsigned char *cls, int r) -> double {
        if (!step_w.empty()) return step_w[(size_t)(cls - table.data()) + r];
        const int c = cls ? cls[r] : 0;
        return c == 1 ? 0.1 : (c == 2 ? w_mixed : 1.0);
    };
    for (int tx = 0; tx < tiles_x; tx++) {
        const unsigned char *cls = classes ? table.data() + (size_t)tx * L : nullptr;
        int r = mid_lo;
        while (r < mid_hi) {
            while (r < mid_hi && cls && cls[r] == 1) r++;
            if (r >= mid_hi) break;
            const int s0 = r;
            int last = r;
            double c = 0.0;
            while (r < mid_hi) {
                if (cls && cls[r] == 1) {
                    int e = r;
                    while (e < mid_hi && cls[e] == 1) e++;
                    if (e - r >= 24 || e == mid_hi) break;
                }
                c += roww(cls, r);
                if (!cls || cls[r] != 1) last = r;
                r++;
            }
            if (twin) st.push_back(stretch{ tx, s0, last + 1, 2 * std::max(1, (last + 1 - s0 + 2 * ST_TWIN_MAXROWS - 1) / (2 * ST_TWIN_MAXROWS)), c });
            else st.push_back(stretch{ tx, s0, last + 1, std::max(1, (last + 1 - s0 + ST_MAXROWS - 1) / ST_MAXROWS), c });
            r = std::max(r, last + 1);
        }
    }
    if (st.empty()) return;
    // a unit of a stretch costs about (its share of the rows) x 8 levels + the fill of its window + the fixed part
    const double fixed = 140.0 + (twin ? 8.0 : 16.0) * 4.5;
    const int inc = twin ? 2 : 1;
    auto per_unit = [&](const stretch &x) { return x.cost * 8.0 / x.k + fixed; };
    int have = 0;
    for (const stretch &x : st) have += x.k;
    if (!twin) want -= n_all;                                // (the send zones' units are part of the round; the twin units of a split launch's
                                                             // interior are a launch of their own, behind the zones' plain units: a round to themselves)
    {
        auto worse = [&](int a, int b) { return per_unit(st[a]) < per_unit(st[b]); };
        std::vector<int> heap(st.size());
        for (size_t i = 0; i < st.size(); i++) heap[i] = (int)i;
        std::make_heap(heap.begin(), heap.end(), worse);
        while (have < want) {
            std::pop_heap(heap.begin(), heap.end(), worse);
            stretch &x = st[heap.back()];
            if ((x.r1 - x.r0) / (x.k + inc) < (twin ? 8 : 12)) break;   // (units shorter than their own fill: the most expensive stretch cannot be helped further)
            x.k += inc; have += inc;
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    }
    for (const stretch &x : st) {
        const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
        int r = x.r0;
        double acc = 0.0;
        if (twin) continue;                                   // (below)
        for (int u = 0; u < x.k; u++) {
            const int u0 = r;
            const double upto = x.cost * (u + 1) / x.k;
            while (r < x.r1 && (u == x.k - 1 || acc + 0.5 * roww(cls, r) <= upto) && r - u0 < ST_MAXROWS) { acc += roww(cls, r); r++; }
            if (u == x.k - 1 && r < x.r1) {                       // (the row limit cut the last unit short: the rest as further units)
                emit(x.tx, cls, u0, r, flipping && (u & 1));
                for (int q = r; q < x.r1; q += ST_MAXROWS) emit(x.tx, cls, q, std::min(q + ST_MAXROWS, x.r1), 0);
                r = x.r1;
                break;
            }
            emit(x.tx, cls, u0, r, flipping && (u & 1));
        }
    }
    if (twin) {
        // Twin units.  A stretch's k units end at rows c[1] .. c[k]; units 2j (southwards) and 2j + 1 (northwards) meet on the
        // seam c[2j + 1].  What a unit costs is not a sum over its rows -- one coast row puts the five steps it is in flight for
        // through the coast formula, a land row makes them tested steps, a unit's steps come in pairs -- so units are priced as
        // the kernel will run them (st_exact_cost), the stretches are packed under a common limit (below), and the boundaries are
        // then moved, up to four rows at a time, towards the dearer of the two units they separate.
        struct cutline { std::vector<int> c; float worst = 0.0f; };
        auto unit_cost = [&](const stretch &x, const std::vector<int> &c, int u) -> float {
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            int a = c[u], b = c[u + 1];
            if (u & 1) { while (cls && b > a && cls[b - 1] == 1) b--; }      // (the far ends are trimmed: emit_twin)
            else { while (cls && a < b && cls[a] == 1) a++; }
            return st_exact_cost(o, x.tx, a, b, !(u & 1), true);
        };
        auto first_cut = [&](const stretch &x, std::vector<int> &c) {
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            c.assign((size_t)x.k + 1, x.r1);
            c[0] = x.r0;
            int r = x.r0;
            double acc = 0.0;
            bool capped = false;
            for (int u = 0; u < x.k; u++) {
                const int u0 = r;
                const double upto = x.cost * (u + 1) / x.k;
                while (r < x.r1 && (u == x.k - 1 || acc + 0.5 * roww(cls, r) <= upto)) { acc += roww(cls, r); r++; }
                c[u + 1] = r;
                capped |= r - u0 > ST_TWIN_MAXROWS;
            }
            if (capped)                                           // (a unit of mostly land grew past the row limit: equal rows instead)
                for (int u = 1; u < x.k; u++) c[u] = x.r0 + (int)((long long)(x.r1 - x.r0) * u / x.k);
        };
        auto refine = [&](const stretch &x, cutline &cl) {
            std::vector<int> &c = cl.c;
            std::vector<float> cost((size_t)x.k);
            for (int u = 0; u < x.k; u++) cost[u] = unit_cost(x, c, u);
            for (int sweep = 0; sweep < 48; sweep++) {
                bool moved = false;
                for (int b = 1; b < x.k; b++) {
                    // (a unit's steps come in pairs, four rows: its cost is a staircase in its rows, so moves of up to four rows are tried)
                    const int dir = cost[b - 1] > cost[b] ? -1 : 1;          // shrink the dearer unit
                    const int old = c[b];
                    float best_max = std::max(cost[b - 1], cost[b]) - 1e-4f, best_a = 0.0f, best_b = 0.0f;
                    int best_nb = old;
                    for (int step = 1; step <= 4; step++) {
                        const int nb = old + dir * step;
                        if (nb - c[b - 1] < 1 || c[b + 1] - nb < 1 || nb - c[b - 1] > ST_TWIN_MAXROWS || c[b + 1] - nb > ST_TWIN_MAXROWS) break;
                        c[b] = nb;
                        const float ca = unit_cost(x, c, b - 1), cb = unit_cost(x, c, b);
                        if (std::max(ca, cb) < best_max) { best_max = std::max(ca, cb); best_a = ca; best_b = cb; best_nb = nb; }
                    }
                    c[b] = best_nb;
                    if (best_nb != old) { cost[b - 1] = best_a; cost[b] = best_b; moved = true; }
                }
                if (!moved) break;
            }
            cl.worst = 0.0f;
            for (int u = 0; u < x.k; u++) cl.worst = std::max(cl.worst, cost[u]);
        };
        std::vector<cutline> cuts(st.size());
        // How many units a stretch gets: the smallest limit T is looked for (bisection) under which the stretches, each packed
        // greedily into units that cost at most T -- a unit grows by four rows, one iteration of the kernel's loop, while it stays
        // under T -- need no more than `want` units together.  (Dealt out by the stretches' summed row weights, the stretches along
        // the coasts came away with too few: their units end on land, and a unit that ends on land has five tested steps
        // whatever its height -- forcing the coast weight to 2.4 in that deal brought the 8-way band from 52.9 to 48.1 us.)
        static const int pack_env = [] { const char *e = getenv("DMG_ST_PACK"); return e ? atoi(e) : 1; }();
        auto cost_rows = [&](const stretch &x, int a, int b, int u) -> float {
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            if (u & 1) { while (cls && b > a && cls[b - 1] == 1) b--; }
            else { while (cls && a < b && cls[a] == 1) a++; }
            return st_exact_cost(o, x.tx, a, b, !(u & 1), true);
        };
        auto pack = [&](const stretch &x, float T, std::vector<int> *c) -> int {
            int r = x.r0, u = 0;
            if (c) { c->clear(); c->push_back(r); }
            while (r < x.r1) {
                int n = std::min(4, x.r1 - r);
                while (r + n < x.r1 && n + 4 <= ST_TWIN_MAXROWS && cost_rows(x, r, std::min(r + n + 4, x.r1), u) <= T) n = std::min(n + 4, x.r1 - r);
                if (x.r1 - (r + n) < 3 && x.r1 - r <= ST_TWIN_MAXROWS) n = x.r1 - r;     // (no sliver at the end)
                r += n; u++;
                if (c) c->push_back(r);
            }
            if (u & 1) { u++; if (c) c->push_back(x.r1); }               // (pairs: an empty unit as the last one's partner)
            return u;
        };
        if (pack_env) {
            float t_lo = 8.0f, t_hi = 8.0f;
            for (const stretch &x : st) t_hi = std::max(t_hi, cost_rows(x, x.r0, std::min(x.r1, x.r0 + ST_TWIN_MAXROWS), 1));
            auto total = [&](float T) { long n = 0; for (const stretch &x : st) n += pack(x, T, nullptr); return n; };
            if (total(t_hi) <= want)
                for (int it = 0; it < 12; it++) {
                    const float mid = 0.5f * (t_lo + t_hi);
                    if (total(mid) <= want) t_hi = mid; else t_lo = mid;
                }
            long have_units = 0;
            for (size_t i = 0; i < st.size(); i++) { st[i].k = pack(st[i], t_hi, &cuts[i].c); refine(st[i], cuts[i]); have_units += st[i].k; }
            // (units grow four rows at a time and come in pairs: the limit that fits leaves some of the round unused -- those go, a
            // pair at a time, to the stretch whose dearest unit is the dearest, while that makes it cheaper)
            std::vector<char> done(st.size(), 0);
            while (have_units + 2 <= want) {
                size_t w = st.size();
                for (size_t i = 0; i < st.size(); i++) if (!done[i] && (w == st.size() || cuts[i].worst > cuts[w].worst)) w = i;
                if (w == st.size()) break;
                stretch x = st[w];
                if ((x.r1 - x.r0) / (x.k + 2) < 4) { done[w] = 1; continue; }
                x.k += 2;
                cutline c2;
                first_cut(x, c2.c); refine(x, c2);
                if (c2.worst < cuts[w].worst - 1e-4f) { st[w] = x; cuts[w] = c2; have_units += 2; }
                else done[w] = 1;
            }
        }
        else for (size_t i = 0; i < st.size(); i++) { first_cut(st[i], cuts[i].c); refine(st[i], cuts[i]); }
        for (size_t i = 0; i < st.size(); i++) {
            const stretch &x = st[i];
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            for (int j = 0; j + 1 < x.k; j += 2) emit_twin(x.tx, cls, cuts[i].c[j], cuts[i].c[j + 1], cuts[i].c[j + 2]);
        }
    }
}

// Segments for the persistent form: every strip cut into pieces of about equal cost (the waves even out the rest among
// themselves).  Runs of 24 or more rows that are land everywhere are left out; a piece starts and ends on a row with water.
static void st_cut_cost(const dmg_ocean *o, int lo, int hi, float target, std::vector<st_unit> &units, double *total)
{
    static const float w_mixed = [] { const char *e = getenv("DMG_ST_WMIXED"); return e ? (float)atof(e) : 1.4f; }();
    static const float w_land = [] { const char *e = getenv("DMG_ST_WLAND"); return e ? (float)atof(e) : 0.15f; }();
    const float fill = 15.0f;                              // a piece of n rows costs about n + 15 (measured: fill and drain of the 8 levels)
    const std::vector<unsigned char> &table = o->rowclass_host;
    const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW, L = o->L;
    units.clear();
    double sum = 0.0;
    for (int tx = 0; tx < tiles_x; tx++) {
        const unsigned char *cls = table.empty() ? nullptr : table.data() + (size_t)tx * L;
        auto cost_of = [&](int r) { const int c = cls ? cls[r] : 0; return c == 1 ? w_land : (c == 2 ? w_mixed : 1.0f); };
        int r = lo;
        while (r < hi) {
            while (r < hi && cls && cls[r] == 1) r++;      // land before the piece
            if (r >= hi) break;
            const int s0 = r;
            float acc = fill;
            int last_water = r;                            // last row of the piece that is not land everywhere
            bool absorb = false;                           // the rest of this stretch is too short for a piece of its own
            while (r < hi) {
                if (cls && cls[r] == 1) {
                    int e = r;
                    while (e < hi && cls[e] == 1) e++;
                    if (e - r >= 24 || e == hi) break;     // a long run of land (or land up to the end): the piece ends before it
                    for (; r < e; r++) acc += w_land;
                    continue;
                }
                acc += cost_of(r);
                last_water = r;
                r++;
                if (!absorb && acc >= target) {
                    // do not leave a sliver behind: if the stretch ends within the next 24 rows, take them along
                    int e = r;
                    bool ends = false;
                    while (e - r < 24) {
                        if (e >= hi) { ends = true; break; }
                        if (cls && cls[e] == 1) {
                            int f = e;
                            while (f < hi && cls[f] == 1) f++;
                            if (f - e >= 24 || f == hi) { ends = true; break; }
                            e = f;
                        } else e++;
                    }
                    if (!ends) break;                      // plenty follows: cut here
                    absorb = true;
                }
            }
            units.push_back(st_unit{ tx, s0, last_water + 1, acc });
            sum += acc;
            r = std::max(r, last_water + 1);
        }
    }
    if (total) *total = sum;
}

// makespan of the units on `slots` waves that each take the next unit when they are done (how the dispatcher hands
// workgroups to free slots), units in descending cost order.
//
// Wide form (two waves per SIMD): the two waves of a SIMD do not share it evenly.  Per-unit timings with the hardware ids
// (DMG_ST_PROF) show the dispatcher filling wave slot 0 of every SIMD with the launch's first 4 x CUs workgroups and slot 1
// with the next, and the SIMD's arbiter serving the older wave first: of two equal units started together the first takes
// 195 us and the second 248 us.  The model: the older (or only) wave of a SIMD advances at 1.25, the younger at 0.75 of the
// even share; when a wave ends, the other one is the older from then on and the next unit of the list starts as the
// younger.  Against a measured 455 / 361 us (no land / survey terrain) it predicts 435 / 357 us, where one speed for all
// slots was 10 % optimistic about the tail and preferred cuts accordingly.
static float st_makespan(std::vector<st_unit> &units, int slots, bool paired = false)
{
    std::sort(units.begin(), units.end(), [](const st_unit &a, const st_unit &b) { return a.cost > b.cost; });
    if (!paired || slots < 2) {
        std::vector<float> heap((size_t)slots, 0.0f);      // min-heap of slot finish times
        auto cmp = [](float a, float b) { return a > b; };
        float span = 0.0f;
        for (const st_unit &u : units) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            heap.back() += u.cost;
            span = std::max(span, heap.back());
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        return span;
    }
    static const float r_old_env = [] { const char *e = getenv("DMG_ST_ROLD"); return e ? (float)atof(e) : 1.25f; }();     // experiments (1.15 .. 1.35 choose the same or equally fast cuts)
    const float r_old = r_old_env, r_young = 2.0f - r_old_env;
    const int nsimd = slots / 2, n = (int)units.size();
    struct simd { double rem[2]; double t; signed char busy[2]; signed char older; };
    std::vector<simd> sm((size_t)nsimd, simd{ { 0.0, 0.0 }, 0.0, { 0, 0 }, 0 });
    typedef std::pair<double, int> ev;                      // (time of the SIMD's next completion, SIMD)
    std::vector<ev> heap;
    heap.reserve((size_t)nsimd + 16);
    auto later = [](const ev &a, const ev &b) { return a.first > b.first; };
    auto schedule = [&](int s) {
        simd &m = sm[s];
        double dt;
        if (m.busy[0] && m.busy[1]) dt = std::min(m.rem[m.older] / r_old, m.rem[1 - m.older] / r_young);
        else if (m.busy[0] || m.busy[1]) dt = m.rem[m.busy[0] ? 0 : 1] / r_old;
        else return;
        heap.push_back(ev(m.t + dt, s));
        std::push_heap(heap.begin(), heap.end(), later);
    };
    int next = 0;
    for (int k = 0; k < 2; k++)
        for (int s = 0; s < nsimd && next < n; s++) { sm[s].rem[k] = units[next++].cost; sm[s].busy[k] = 1; }
    for (int s = 0; s < nsimd; s++) schedule(s);
    float span = 0.0f;
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), later);
        const ev e = heap.back();
        heap.pop_back();
        simd &m = sm[e.second];
        const double dt = e.first - m.t;
        if (m.busy[0] && m.busy[1]) { m.rem[m.older] -= dt * r_old; m.rem[1 - m.older] -= dt * r_young; }
        else if (m.busy[0] || m.busy[1]) m.rem[m.busy[0] ? 0 : 1] -= dt * r_old;
        m.t = e.first;
        for (int k = 0; k < 2; k++)
            if (m.busy[k] && m.rem[k] <= 1e-6) {
                m.busy[k] = 0;
                span = std::max(span, (float)m.t);
                if (m.busy[1 - k]) m.older = (signed char)(1 - k);
                if (next < n) {
                    m.rem[k] = units[next++].cost; m.busy[k] = 1;
                    if (!m.busy[1 - k]) m.older = (signed char)k;
                }
            }
        schedule(e.second);
    }
    return span;
}

// Where the workgroups of a one-round launch of twin units run, and next to whom.  Per-unit timings with the hardware ids
// (DMG_ST_PROF, profiles/r04_twin_units.txt; 256 of 256 CUs alike): workgroup i of a launch of 128-thread workgroups runs on
// XCD (i + const) mod 8 and on the CU that also gets workgroups i + 256, i + 512, i + 768 (four fit: LDS), and the CU deals their
// eight waves w0 .. w7 (two per workgroup, in order) to its SIMDs as  S, S+1 | S+1, S+2 | S+2, S+3 | S+3, S  -- so w1 and w2
// share a SIMD, w3 and w4, w5 and w6, w7 and w0, the first of each couple being the older wave there, the one the SIMD's
// arbiter serves first.  A SIMD is done when both its waves are: a launch of one round lasts as long as the SIMD with the
// most work, and with the workgroups in longest-first order that is the one whose older wave came from the dearest quarter
// of the list and its younger wave from the next (predicted sums 34.6 .. 45.0 around a median of 38.0, and the twelve units
// that ended last all sat on such SIMDs).  Here the list is ORDERED for that hardware: every XCD gets the pairs of a contiguous
// range of strips (neighbouring strips share halo columns in its L2, as before); within it the dearest pairs become the CUs'
// first workgroups, the cheapest their fourth, matched dearest with cheapest, the second and third likewise; and within a CU the
// order of the two units inside each workgroup (which of them is the first wave) is chosen, all 16 ways tried, so that the
// four SIMDs' sums are as equal as they can be, the older wave the dearer one where that costs nothing.  If the dispatcher
// ever deals differently the order is merely arbitrary -- no worse than any other.  Lists of fewer than four workgroups per CU
// are padded with workgroups that end at once, so that position i + 256 k stays on CU i.
static void st_place_twins(std::vector<st_unit> &units, int num_cu, int tiles_x)
{
    const int C = std::max(num_cu, 8) / 8 * 8, per_x = C / 8;          // CUs, CUs per XCD
    const int n = (int)units.size();
    const int R = (n + C - 1) / C;                                      // workgroups per CU
    if (n == 0 || R > 4) return;
    std::vector<st_unit> sorted = units;
    std::sort(sorted.begin(), sorted.end(), [](const st_unit &a, const st_unit &b) { return a.tx != b.tx ? a.tx < b.tx : a.r0 < b.r0; });
    st_unit dummy{ 0, 0, 0, 0.0f, 0 };
    dummy.rlo = 0; dummy.dummy = 1;
    std::vector<st_unit> out((size_t)R * C, dummy);
    double couple_lo = 1e30, couple_hi = 0.0;
    for (int x = 0; x < 8; x++) {
        // this XCD's pairs: a contiguous eighth of the list in strip order
        const int g0 = (int)((long long)n * x / 8), g1 = (int)((long long)n * (x + 1) / 8);
        std::vector<st_unit> g(sorted.begin() + g0, sorted.begin() + g1);
        const int m = (int)g.size();
        if (!m) continue;
        // Every workgroup's dearer unit is its SECOND wave: the couples of a CU are then (dearer unit of workgroup r, cheaper unit of
        // workgroup r + 1), cyclically, the dearer one the older wave of its SIMD.  The CU's workgroups are chosen as a cycle whose
        // couples all come close to the mean sum: start from the unused pair with the dearest unit, follow it with the pair whose
        // cheaper unit complements that best, and so on; the last one also has to suit the first.
        auto hi = [](const st_unit &u) { return std::max(u.cost_n, u.cost_s); };
        auto lo = [](const st_unit &u) { return std::min(u.cost_n, u.cost_s); };
        double target = 0.0;
        for (const st_unit &u : g) target += u.cost_n + u.cost_s;
        target /= m;                                                    // the mean couple
        std::vector<char> used((size_t)m, 0);
        int left = m;
        for (int t = 0; t < per_x && left > 0; t++) {
            const int len = std::min(left, (left + (per_x - t) - 1) / (per_x - t));      // this CU's share of what is left (at most R)
            int cyc[4] = { -1, -1, -1, -1 };
            for (int r = 0; r < len; r++) {
                int pick = -1;
                double best_d = 1e30;
                for (int i = 0; i < m; i++) {
                    if (used[i]) continue;
                    double dd;
                    if (r == 0) dd = -hi(g[i]);                        // the dearest unit left opens the cycle
                    else {
                        dd = std::fabs(hi(g[cyc[r - 1]]) + lo(g[i]) - target);
                        if (r == len - 1) dd = std::max(dd, std::fabs(hi(g[i]) + lo(g[cyc[0]]) - target));
                    }
                    if (dd < best_d) { best_d = dd; pick = i; }
                }
                cyc[r] = pick; used[pick] = 1; left--;
            }
            for (int r = 0; r < len; r++) {
                st_unit u = g[cyc[r]];
                u.south_first = u.cost_s < u.cost_n ? 1 : 0;           // the cheaper unit first
                out[(size_t)r * C + 8 * t + x] = u;
                const double couple = hi(g[cyc[r]]) + lo(g[cyc[(r + 1) % len]]);
                couple_lo = std::min(couple_lo, couple); couple_hi = std::max(couple_hi, couple);
            }
        }
    }
    // (trailing dummies are dropped: nothing follows them)
    while (!out.empty() && out.back().dummy) out.pop_back();
    static const bool dbg = getenv("DMG_ST_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[stream]   %d pairs placed on %d CUs, %d per CU: the SIMDs' predicted sums %.1f .. %.1f steps\n", n, C, R, couple_lo, couple_hi);
    units.swap(out);
}

// The unit list of a launch geometry.  A launch has only 2 x 4 x CUs wave slots and its units all take about the same
// time, so the chip works in "rounds": 2.49 rounds of 96-row units cost as much as 3.  The unit height is therefore chosen
// per geometry (and terrain) by simulating the dispatch for every candidate height -- taller units recompute less
// (7/n), but the last round must be full -- and the units are issued longest first.
// part 1 / 2: only the units that do / do not produce rows of [lo, zone_lo) or [zone_hi, hi) (the two halves of a launch
// split around a halo exchange).
static thread_local bool st_units_failed = false;         // the last st_units call of this thread failed with a message of its own
static const dmg_ocean::tile_list *st_units(dmg_ocean *o, int variant, int lo, int hi, int part, int zone_lo, int zone_hi, int seg)
{
    st_units_failed = false;
    auto key = std::make_tuple(lo, hi, (variant == 2 ? -5000 : (variant ? -3000 : -1000)) - seg, part, zone_lo, zone_hi);    // "rows per wave" <= -1000 = stream units (seg 0 = auto)
    auto it = o->tb_lists.find(key);
    if (it != o->tb_lists.end()) return &it->second;
    const int slots = std::max(o->ctx->num_cu, 1) * (variant == 1 ? 20 : 8);     // five (96 VGPRs) / two waves per SIMD
    std::vector<st_unit> units, best;
    int p_waves = 0;
    int cut_seg = 0, cut_seg2 = 0, cut_split = 0;         // the cut this list was made with (st_cut's seg, seg2, split)
    int cut_twin = 0;                                     // ... a balanced cut into twin units
    if (variant == 2) {
        // persistent form: about one segment per wave (the waves balance the rest among themselves); a strip's rows that
        // are land everywhere at either end of a cut are trimmed as for the units
        double total = 0.0;
        st_cut_cost(o, lo, hi, 1e30f, units, &total);      // uncut: one piece per stretch of water, for the total
        p_waves = (int)std::max(1.0, std::min((double)std::max(o->ctx->num_cu, 1) * 8, total / 32.0));
        // every further cut adds a fill: solve target = (total + 15 (pieces - stretches)) / waves for pieces ~ waves
        float target = (float)((total + 15.0 * std::max(0, p_waves - (int)units.size())) / p_waves);
        if (seg > 0) target = (float)seg;
        st_cut_cost(o, lo, hi, std::max(target, 39.0f), best, &total);
        (void)st_makespan(best, p_waves);                  // (sorts: longest first)
        static const bool dbg = getenv("DMG_ST_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[stream] rows [%d,%d): persistent, %d waves, %zu segments of cost %.0f (total %.0f, longest %.0f, shortest %.0f)\n", lo, hi, p_waves,
                         best.size(), target, total, best.empty() ? 0.0f : best.front().cost, best.empty() ? 0.0f : best.back().cost);
    }
    else if (part) {
        // The two halves of a split launch are halves of ONE cut: the one the whole launch over these rows uses (chosen
        // with every unit in play), filtered by the edge predicate.  (Each half choosing its own unit heights left the
        // rows between the two choices to neither of them.)
        const dmg_ocean::tile_list *whole = st_units(o, variant, lo, hi, 0, 0, 0, seg);
        if (!whole) return nullptr;
        cut_seg = whole->cut_seg; cut_seg2 = whole->cut_seg2; cut_split = whole->cut_split; cut_twin = whole->cut_twin;
        if (cut_seg == 0) { cut_seg = variant == 1 ? STN_MAXROWS : ST_MAXROWS; cut_seg2 = 0; cut_split = hi; }   // (a launch with no live unit)
        static const int edge_seg = [] { const char *e = getenv("DMG_ST_EDGE_SEG"); const int v = e ? atoi(e) : 4; return v >= 2 && v <= ST_MAXROWS ? v : 0; }();
        auto cut_as_whole = [&](int which, std::vector<st_unit> &out) {      // (cut_seg < 0: the balanced cut, -cut_seg units in all)
            if (cut_seg < 0) st_cut_balanced(o, lo, hi, -cut_seg, which, zone_lo, zone_hi, out, edge_seg, cut_twin != 0);
            else st_cut(o, variant, lo, hi, cut_seg, which, zone_lo, zone_hi, out, cut_seg2, cut_split, edge_seg);
        };
        cut_as_whole(part, best);
        (void)st_makespan(best, slots);
        auto units_of = [](const std::vector<st_unit> &v) { size_t n = 0; for (const st_unit &u : v) n += u.dummy ? 0 : (u.rlo >= 0 ? 2 : 1); return n; };
        // both halves built: together they must be the units of ONE cut of the launch's rows, each exactly once
        auto other = o->tb_lists.find(std::make_tuple(lo, hi, std::get<2>(key), 3 - part, zone_lo, zone_hi));
        if (other != o->tb_lists.end()) {
            std::vector<st_unit> all;
            cut_as_whole(0, all);
            long long rows_all = 0, rows_parts = 0;
            for (const st_unit &u : all) rows_all += u.r1 - u.first();
            for (const st_unit &u : best) rows_parts += u.r1 - u.first();
            if (other->second.n_cut_units + units_of(best) != units_of(all) || other->second.cut_rows + rows_parts != rows_all) {
                st_units_failed = true; dmg_fail(DMG_ESTATE, "split launch over rows [%d, %d): %zu + %zu units in the two parts, %zu in the cut they are halves of", lo, hi,
                         other->second.n_cut_units, units_of(best), units_of(all));
                return nullptr;
            }
        }
    }
    else if (seg > 0) { st_cut(o, variant, lo, hi, seg, part, zone_lo, zone_hi, best); (void)st_makespan(best, slots); cut_seg = seg; cut_split = hi; }
    else {
        float best_span = 1e30f;
        int best_seg = 0, best_seg2 = 0, best_split = 0;
        // what a cut is judged by: alpha * (makespan of the simulated dispatch) + (1 - alpha) * (its work / slots).  The chip's
        // throughput is shared by the two waves of a SIMD -- when a slot idles, its neighbour speeds up -- so an idle tail costs
        // less than its length and the work (taller units stream fewer fill rows) counts for more than the makespan alone says
        static const float alpha_env = [] { const char *e = getenv("DMG_ST_ALPHA"); return e ? (float)atof(e) : -1.0f; }();
        const float alpha = alpha_env >= 0.0f ? alpha_env : (seg == -2 ? 0.5f : 1.0f);                  // seg -2: the second automatic cut
        auto judge = [&](std::vector<st_unit> &u) {
            const float span = st_makespan(u, slots, variant == 0);
            double total = 0;
            for (const st_unit &x : u) total += x.cost;
            return alpha * span + (1.0f - alpha) * (float)(total / slots);
        };
        // (every height up to 48 rows: a band of a thousand rows has ~25 rows per wave slot, and one row more or less per unit
        // decides whether the launch's single round fills the slots or spills into a second)
        static const int fine_env = [] { const char *e = getenv("DMG_ST_FINE"); return e ? atoi(e) : 1; }();
        for (int cand = fine_env ? 20 : 24; cand <= (variant == 1 ? STN_MAXROWS : ST_MAXROWS); cand += (fine_env && cand < 48) ? 1 : 4) {
            st_cut(o, variant, lo, hi, cand, part, zone_lo, zone_hi, units);
            float span = judge(units);
            if (span < best_span) { best_span = span; best.swap(units); best_seg = cand; best_seg2 = 0; best_split = hi; }
            // two tiers: the top eighth .. half of the rows in units a half, a third or a quarter as tall
            for (int frac = 1; frac <= 4 && cand >= 48; frac++)
                for (int div = 2; div <= 4; div++) {
                    const int split = hi - ((hi - lo) * frac / 8 / cand) * cand, seg2 = ((cand / div + 3) / 4) * 4;
                    if (split <= lo || split >= hi || seg2 < 12) continue;
                    st_cut(o, variant, lo, hi, cand, part, zone_lo, zone_hi, units, seg2, split);
                    span = judge(units);
                    if (span < best_span) { best_span = span; best.swap(units); best_seg = cand; best_seg2 = seg2; best_split = split; }
                }
        }
        // ... and the strip-by-strip cut into units of equal cost (st_cut_balanced), at one, two and three rounds' worth of units
        static const int bal_env = [] { const char *e = getenv("DMG_ST_BALANCED"); return e ? atoi(e) : 1; }();     // how many rounds' worth at most (0: off)
        // (only on launches of at most ST_BALANCED_ROWS rows whose best fixed-height cut is itself about one round -- the band of an
        // 8-way split.  There it is worth 5-9 % (16384 x 1024: 65.0 -> 59.5 / 61.6 us on two boxes; the middle band of the 8-way split
        // 62.3 -> 59.3 us).  On taller launches the simulation's preference is not the chip's: 16384 x 2048 in one round of 2048
        // equal-cost units takes 94 us against 87 for 1981 units of 52 rows -- strips whose units no longer cover the same rows at
        // the same time stop sharing their halo columns in the L2 -- although the 4-way middle band gains 4.5 %;
        // profiles/r04_balanced_cut.txt)
        static const int bal_rows = [] { const char *e = getenv("DMG_ST_BALANCED_ROWS"); return e ? atoi(e) : 1200; }();
        if (bal_env && variant == 0 && !o->rowclass_host.empty() && (int)best.size() <= slots + slots / 10 && hi - lo <= bal_rows)
            for (int eighths = 6; eighths <= 8 * bal_env; eighths += (eighths < 8 ? 1 : 8)) {      // 3/4, 7/8, 1 (, 2, 3) rounds' worth of units
                const int want = slots * eighths / 8;
                st_cut_balanced(o, lo, hi, want, part, zone_lo, zone_hi, units);
                if (units.empty()) continue;
                const float span = judge(units);
                static const bool dbgc = getenv("DMG_ST_DEBUG") != nullptr;
                if (dbgc) fprintf(stderr, "[stream]   balanced cut of %d units: predicted span %.0f (best so far %.0f)\n", want, span, best_span);
                if (span < best_span) { best_span = span; best.swap(units); best_seg = -want; best_seg2 = 0; best_split = hi; }
            }
        // ... and into twin units (pairs that start on a common seam and need no halo between them: k_jacobi_stream_twin), one round's worth
        // (read whenever a list is built, not once per process: bench.py switches the twin units off and tries again if every leg
        // of a first multi-GPU run fails with them)
        const int twin_env = [] { const char *e = getenv("DMG_ST_TWIN"); return e ? atoi(e) : ST_TWIN_DEFAULT; }();
        // (up to the band of a 2-way split.  16384 x 2048: one round of twin units 73.1 us against 81.6 on the survey terrain, 93.9 against
        // 114.2 without land; 16384 x 4096: 144.2-145.6 against 149.9 / 187.3 against 188.5.  At 16384 x 8192 they LOSE to 1.3 rounds of
        // 200-row plain units, as two or three rounds of short pairs (292-296 against 273.9 us, 395 against 362.6 without land) and as
        // one round of pairs of up to 232 rows (300.2 against 279.9 us): a launch of one round lasts as long as its slowest SIMD, 1.2
        // medians, where the plain cut's short units fill the slots its tall ones leave -- profiles/r04_twin_units.txt)
        static const int twin_rows = [] { const char *e = getenv("DMG_ST_TWIN_ROWS"); return e ? atoi(e) : 4400; }();
        if (twin_env && bal_env && variant == 0 && !o->rowclass_host.empty() && hi - lo <= twin_rows) {
            // (the twin units' costs are st_exact_cost's: the cut they are compared with is priced the same way)
            std::vector<st_unit> ref = best;
            for (st_unit &u : ref) u.cost = 16.0f * st_exact_cost(o, u.tx, u.r0, u.r1, u.flip != 0, false);
            const float ref_span = ref.empty() ? 1e30f : st_makespan(ref, slots, true);
            float twin_span = 1e30f;
            std::vector<st_unit> twin_best;
            int twin_want = 0;
            // one round's worth of units (DMG_ST_TWIN_ROUNDS=3: also one and a half, two and three rounds' worth -- measured, never better)
            static const int twin_rounds = [] { const char *e = getenv("DMG_ST_TWIN_ROUNDS"); return e ? atoi(e) : 1; }();
            for (int eighths = 8; eighths <= 8 * twin_rounds; eighths += (eighths < 16 ? 4 : 8)) {
                const int want = slots * eighths / 8;
                const auto t_cut0 = std::chrono::steady_clock::now();
                st_cut_balanced(o, lo, hi, want, part, zone_lo, zone_hi, units, 0, true);
                const double cut_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cut0).count();
                if (units.empty()) continue;
                const float span = st_makespan(units, slots / 2);          // (a pair holds two of the slots)
                static const int dbgc = [] { const char *e = getenv("DMG_ST_DEBUG"); return e ? atoi(e) : 0; }();
                if (dbgc) {
                    double sum = 0; float lo_c = 1e30f;
                    for (const st_unit &u : units) { sum += u.cost; lo_c = std::min(lo_c, u.cost); }
                    fprintf(stderr, "[stream]   balanced cut into %zu pairs of twin units (%d asked for, %.0f ms on the host): predicted span %.0f (dearest pair %.0f, mean %.0f, cheapest %.0f); the best plain cut priced alike: %.0f\n",
                            units.size(), want / 2, cut_ms, span, units.front().cost, sum / units.size(), lo_c, ref_span);
                    for (size_t i = 0; i < units.size() && i < 12 && dbgc > 1; i++) {
                        const st_unit &u = units[i];
                        const unsigned char *cls = o->rowclass_host.data() + (size_t)u.tx * o->L;
                        const st_features fn = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, u.r0, u.r1, false, true);
                        const st_features fs = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, o->L - u.r0, o->L - u.rlo, true, true);
                        fprintf(stderr, "[stream]     pair %zu: strip %d rows [%d, %d | %d) cost %.0f; north: %d fast %d coast %d tested (%d/%d/%d) fill (%d/%d/%d) = %.1f; south: %d fast %d coast %d tested (%d/%d/%d) fill (%d/%d/%d) = %.1f\n",
                                i, u.tx, u.rlo, u.r0, u.r1, u.cost, fn.fast, fn.coast, fn.tested, fn.t_open, fn.t_mixed, fn.t_land, fn.f_open, fn.f_mixed, fn.f_land,
                                st_exact_cost(o, u.tx, u.r0, u.r1, false, true), fs.fast, fs.coast, fs.tested, fs.t_open, fs.t_mixed, fs.t_land, fs.f_open, fs.f_mixed, fs.f_land,
                                st_exact_cost(o, u.tx, u.rlo, u.r0, true, true));
                    }
                }
                if (span < twin_span) { twin_span = span; twin_best.swap(units); twin_want = want; }
            }
            if (!twin_best.empty() && (twin_span < ref_span || twin_env > 1)) { best_span = twin_span; best.swap(twin_best); best_seg = -twin_want; best_seg2 = 0; best_split = hi; cut_twin = 1; }
        }
        (void)st_makespan(best, cut_twin ? slots / 2 : slots);      // (longest first)
        cut_seg = best_seg; cut_seg2 = best_seg2; cut_split = best_split;
        static const bool dbg = getenv("DMG_ST_DEBUG") != nullptr;
        if (dbg) {
            double total = 0;
            for (const st_unit &u : best) total += u.cost;
            fprintf(stderr, "[stream] rows [%d,%d) part %d: %zu units, %s %d rows up to row %d then %d; predicted span %.0f, mean load %.0f (%.1f %% of the slots' time)\n",
                    lo, hi, part, best.size(), best_seg < 0 ? "balanced cut of" : "", best_seg < 0 ? -best_seg : best_seg, best_split, best_seg2, best_span, total / slots,
                    100.0 * total / slots / best_span);
            // which path the steps of these units take (host classes: 0 open, 1 land, 2 mixed), and how many of the rows that go
            // through the coast formula in an all-coast step are open water
            const std::vector<unsigned char> &table = variant ? o->rowclass_n_host : o->rowclass_host;
            if (!table.empty() && !cut_twin) {
                long long n1 = 0, n2 = 0, n0 = 0, rows2 = 0, open2 = 0;
                for (const st_unit &u : best) {
                    const unsigned char *cls = table.data() + (size_t)u.tx * o->L;
                    const int e0 = u.r0 - TB_T, nst = (((u.r1 - u.r0) + 2 * TB_T + 3) / 4) * 2;
                    for (int t = TB_T; t < nst; t++) {
                        bool land = false, mix = false; int nopen = 0;
                        for (int k = 0; k < 10; k++) {
                            const int r = e0 + 2 * t + 1 - k;
                            const int c = (r >= 0 && r < o->L) ? cls[r] : 2;
                            land |= c == 1; mix |= c == 2; nopen += c == 0;
                        }
                        if (land) n0++; else if (mix) { n2++; rows2 += 10; open2 += nopen; } else n1++;
                    }
                }
                fprintf(stderr, "[stream]   steps past the fill: %.1f %% open (no test), %.1f %% all through the coast formula, %.1f %% tested; of the rows in flight in coast steps %.1f %% are open water\n",
                        100.0 * n1 / (n1 + n2 + n0 + 1e-9), 100.0 * n2 / (n1 + n2 + n0 + 1e-9), 100.0 * n0 / (n1 + n2 + n0 + 1e-9), 100.0 * open2 / (rows2 + 1e-9));
            }
        }
    }
    {
        // Workgroup i of a launch runs on XCD (i + const) mod 8 (per-unit hardware ids, DMG_ST_PROF), each XCD behind its own L2.
        // Among units of equal cost -- whose order longest-first leaves open -- every XCD gets a contiguous range of strips,
        // row block by row block: the 16 halo columns two neighbouring strips share are then read through one L2 instead of
        // two (+0.7 % on the survey terrain, +1.0 % without land in a same-box A/B; wider cost buckets disturb longest-first
        // and lose).  DMG_ST_XCC=0 keeps the sorted order.
        static const int xcc_mode = [] { const char *e = getenv("DMG_ST_XCC"); return e ? atoi(e) : 1; }();
        static const int place_mode = [] { const char *e = getenv("DMG_ST_PLACE"); return e ? atoi(e) : 1; }();
        const bool twins = !best.empty() && best.front().rlo >= 0;
        if (twins && place_mode && (int)best.size() <= slots / 2) st_place_twins(best, o->ctx->num_cu, (o->grid.W + TB_OW - 1) / TB_OW);
        else if (xcc_mode && variant == 0 && !best.empty()) {
            const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW;
            const float top = best.front().cost;
            std::vector<st_unit> out;
            out.reserve(best.size());
            size_t i0 = 0;
            while (i0 < best.size()) {
                size_t i1 = i0;
                const float floor_cost = best[i0].cost - 1e-4f * top;                     // a run of equal costs
                while (i1 < best.size() && best[i1].cost >= floor_cost) i1++;
                std::vector<st_unit> lists[8];
                for (size_t i = i0; i < i1; i++) lists[std::min(7, best[i].tx * 8 / std::max(tiles_x, 1))].push_back(best[i]);
                for (int x = 0; x < 8; x++)
                    std::sort(lists[x].begin(), lists[x].end(), [](const st_unit &a, const st_unit &b) { return a.r0 != b.r0 ? a.r0 < b.r0 : a.tx < b.tx; });
                size_t pos[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                for (size_t done = 0; done < i1 - i0;)
                    for (int x = 0; x < 8; x++) {
                        // (keep the slot of XCD x even when its list has run dry: take from the longest remaining one)
                        int y = x;
                        if (pos[y] >= lists[y].size()) { size_t bestn = 0; for (int q = 0; q < 8; q++) if (lists[q].size() - pos[q] > bestn) { bestn = lists[q].size() - pos[q]; y = q; } if (!bestn) break; }
                        out.push_back(lists[y][pos[y]++]);
                        done++;
                        if (done == i1 - i0) break;
                    }
                i0 = i1;
            }
            best.swap(out);
        }
    }
    std::vector<int> flat;
    flat.reserve(best.size() * 4);
    // a unit is an open-water unit (bit 0 of its fourth word; st_open_unit) if every row of its window, and every row its
    // steps request past the window, is plain open water with exact divisor pairs inside the grid and the stored band
    static const bool open_units = [] { const char *e = getenv("DMG_ST_OPEN_UNITS"); return e && atoi(e) != 0; }();     // (measured: no gain; off by default)
    const std::vector<unsigned char> &ctab = o->rowclass_host;
    int n_open = 0, n_flip = 0, n_twin = 0;
    for (const st_unit &u : best) n_twin += u.rlo >= 0;
    if (n_twin && getenv("DMG_ST_TWIN_FAIL")) {            // tests: a launch that would run twin units fails instead (bench.py's fall-back to plain units)
        st_units_failed = true; dmg_fail(DMG_ESTATE, "twin units: injected failure (DMG_ST_TWIN_FAIL)");
        return nullptr;
    }
    if (n_twin && n_twin != (int)best.size()) { st_units_failed = true; dmg_fail(DMG_ESTATE, "a unit list mixes twin units with plain ones"); return nullptr; }
    for (const st_unit &u : best) {
        int flag = 0;
        if (u.rlo >= 0) {                                  // the unit above the seam and the one below it in the mirrored view's rows
            if (u.r1 - u.r0 > ST_TWIN_MAXROWS || u.r0 - u.rlo > ST_TWIN_MAXROWS) { st_units_failed = true; dmg_fail(DMG_ESTATE, "a twin unit of more than %d rows", ST_TWIN_MAXROWS); return nullptr; }
            // (bit 4: the units of a split launch's edge half -- the exchange waits for them: they keep a raised priority)
            const int urgent = part == 1 ? 16 : 0;
            const int north[4] = { u.tx, u.r0, u.r1, (u.dummy ? 4 | 8 : 4) | urgent }, south[4] = { u.tx, o->L - u.r0, o->L - u.rlo, (u.dummy ? 4 | 2 | 8 : 4 | 2) | urgent };
            flat.insert(flat.end(), u.south_first ? south : north, (u.south_first ? south : north) + 4);
            flat.insert(flat.end(), u.south_first ? north : south, (u.south_first ? north : south) + 4);
            n_flip++;
            continue;
        }
        if (open_units && variant == 0 && !ctab.empty()) {
            const int e0 = u.r0 - TB_T, nst = (((u.r1 - u.r0) + 2 * TB_T + 3) / 4) * 2, e_last = e0 + 2 * nst + 5;
            const int y_first = o->y0 - o->gs;
            bool ok = e0 >= 0 && e_last < o->L && y_first + e0 >= 0 && y_first + e_last < o->grid.H;
            const unsigned char *cls = ctab.data() + (size_t)u.tx * o->L;
            for (int r = e0; ok && r <= e0 + 2 * nst + 1; r++) ok = cls[r] == 0 && (o->rowbad_host.empty() || !o->rowbad_host[r]);
            flag = ok ? 1 : 0;
        }
        n_open += flag;
        if (u.flip && !flag) { flat.push_back(u.tx); flat.push_back(o->L - u.r1); flat.push_back(o->L - u.r0); flat.push_back(2); n_flip++; }   // the mirrored view's rows
        else { flat.push_back(u.tx); flat.push_back(u.r0); flat.push_back(u.r1); flat.push_back(flag); }
    }
    {
        static const bool dbg = getenv("DMG_ST_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[stream] rows [%d,%d) part %d: %d of %zu units are open-water units, %d stream southwards\n", lo, hi, part, n_open, best.size(), n_flip);
    }
    dmg_ocean::tile_list tl;
    tl.n = (int)best.size();
    tl.twin = n_twin ? 1 : 0;
    tl.n_units = (size_t)(n_twin ? 2 * tl.n : tl.n);
    tl.n_cut_units = 0;
    for (const st_unit &u : best) tl.n_cut_units += u.dummy ? 0 : (u.rlo >= 0 ? 2 : 1);
    tl.n_flip = n_flip;
    tl.cut_seg = cut_seg; tl.cut_seg2 = cut_seg2; tl.cut_split = cut_split; tl.cut_twin = cut_twin;
    tl.cut_rows = 0;
    for (const st_unit &u : best) tl.cut_rows += u.r1 - u.first();
    if (o->lists_on_host) { tl.host = flat; return &(o->tb_lists[key] = tl); }       // (dmg_stream_units_probe: no device)
    if (tl.n && hipMalloc((void **)&tl.d, flat.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (tl.n && hipMemcpy(tl.d, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (variant == 2 && tl.n) {
        tl.waves = p_waves;
        tl.cap = ((tl.n + p_waves + 511) / 512) * 512;   // initial segments + about one steal per wave (a thief that finds the table full keeps its piece to itself)
        std::vector<unsigned long long> w0((size_t)tl.n);
        std::vector<unsigned> c0((size_t)tl.n);
        for (int i = 0; i < tl.n; i++) {
            w0[i] = stp_word(best[i].tx, best[i].r0, best[i].r1);
            c0[i] = (unsigned)(best[i].r0 - TB_T + STP_BIAS);
        }
        const size_t total = (size_t)tl.cap + tl.waves;
        if (hipMalloc((void **)&tl.seg_word0, w0.size() * 8) != hipSuccess || hipMalloc((void **)&tl.seg_cur0, c0.size() * 4) != hipSuccess ||
            hipMalloc((void **)&tl.seg_word, total * 8) != hipSuccess || hipMalloc((void **)&tl.seg_cur, total * 4) != hipSuccess ||
            hipMalloc((void **)&tl.seg_ctl, 2 * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemcpy(tl.seg_word0, w0.data(), w0.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(tl.seg_cur0, c0.data(), c0.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    }
    return &(o->tb_lists[key] = tl);
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
        if (getenv("DMG_ST_DEBUG")) fprintf(stderr, "[stream] no unit list for rows [%d,%d) part %d: %s\n", lo, hi, part, st_units_failed ? dmg_last_error() : "allocation failed");
        return st_units_failed ? DMG_ESTATE : dmg_fail(DMG_ENOMEM, "unit list allocation failed");     // (st_units has said what failed)
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

// Test hook (include/demiurge_hip.h): the unit list the streaming kernel would be launched with, from a row-class table, on
// the host alone -- the cut, the halves of a split launch and the placement are plain host code, and the multi-GPU paths that
// depend on them cannot run on a one-GPU box.
extern "C" int dmg_stream_units_probe(int W, int H, int y0, int y1, int gs, int gn, const unsigned char *rowclass, int num_cu,
                                      int lo, int hi, int part, int zone_lo, int zone_hi, int *out, size_t cap, size_t *n_entries, int *twin)
{
    if (W < TB_EW || H < 1 || y0 < 0 || y1 <= y0 || y1 > H || gs < 0 || gn < 0 || !rowclass || num_cu < 1 || !n_entries)
        return dmg_fail(DMG_EINVAL, "dmg_stream_units_probe: bad arguments");
    dmg_ctx ctx;
    ctx.num_cu = num_cu;
    dmg_ocean o;
    o.ctx = &ctx;
    o.grid.W = W; o.grid.H = H;
    o.y0 = y0; o.y1 = y1; o.gs = gs; o.gn = gn;
    o.L = gs + (y1 - y0) + gn;
    o.lists_on_host = true;
    if (lo < 0 || hi > o.L || lo >= hi) return dmg_fail(DMG_EINVAL, "dmg_stream_units_probe: rows [%d, %d) outside the band's %d", lo, hi, o.L);
    const int tiles_x = (W + TB_OW - 1) / TB_OW;
    o.rowclass_host.assign(rowclass, rowclass + (size_t)tiles_x * o.L);
    const dmg_ocean::tile_list *tl = st_units(&o, 0, lo, hi, part, zone_lo, zone_hi, 0);
    if (!tl) return st_units_failed ? DMG_ESTATE : dmg_fail(DMG_ENOMEM, "unit list allocation failed");
    *n_entries = tl->host.size() / 4;
    if (twin) *twin = tl->twin;
    if (out && tl->host.size() <= cap * 4) memcpy(out, tl->host.data(), tl->host.size() * sizeof(int));
    else if (out) return dmg_fail(DMG_EINVAL, "dmg_stream_units_probe: %zu entries, room for %zu", tl->host.size() / 4, cap);
    return DMG_OK;
}
