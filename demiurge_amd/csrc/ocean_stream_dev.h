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
// levels without a test or a branch; one ring read per row and level.  Two forms share the code: units (the default for
// the full grid) and twin units (round 4, the default for launches of up to 4400 rows -- the bands of a split grid: two waves
// that start on the two sides of a seam and need no halo between them, k_jacobi_stream_twin); each in three arithmetics
// (DV: Markstein divisions, exact two-operation divisions -- the default -- and round 6's tolerance rendering with fused row
// coefficients).  (Narrow units and persistent waves with work stealing were retired in round 5: profiles/NOTEBOOK.md.)
//
// Canvas topology exactly as in ocean_jacobi_tb.hip: columns wrap, rows beyond a pole are the same rows mirrored and
// shifted by W/2 (offset(), src/Shader.h:86-95), rows beyond a non-pole edge are plain mirrors with the clamp bit
// cleared.  Arithmetic: jacobi_value() of ocean_internal.h, the same operations in the same order as every other Jacobi kernel.
#pragma once
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
#ifndef ST_RING
#define ST_RING 12                // ring slots (rows): 10 rows are in flight between level 0 and level 8, two arrive
#endif
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
    int nseg0;                      // the deep form's prototype (ocean_jacobi_deep.hip, experiments): units in the list
    unsigned *deep_err;             // the deep form: a word the launch ORs 1 into when a wave gave up waiting for its partner (never, unless the protocol is broken)
    unsigned long long *prof;       // experiments only (DMG_ST_PROF): {start, end} of every unit on the 100 MHz s_memrealtime clock, or null
    unsigned long long *clk;        // experiments only (-DST_EXPERIMENTS, with DMG_ST_PROF): s_memtime (the shader clock's counter) at every unit's start and end, beside prof's s_memrealtime
    unsigned long long *stepprof;   // experiments only (-DST_EXPERIMENTS, DMG_ST_STEPPROF): s_memrealtime at the top of the first 64 steps of every unit
    int dbg;                        // experiments only (DMG_ST_DBG bits): 1 = every unit re-reads 64 rows (cache hits), 2 = no stores, 4 = plain instead of nontemporal stores;
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
// DV = 3, the tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE; oracle/dmg_oracle.c form 3): the statement is linear in its operands,
// fc = a (pW + pE) + b (pS + pN) - c bC with a = 1/(pw2.x Beta), b = 1/(pw2.y Beta), c = 1/Beta taken from the reference's own fp32
// row constants in fp64.  a = ah + al, b = bh + bl with ah + bh = 0.5 exactly and e = RN(al + bl): no systematic error in the gain
// of a smooth field, which 5000 sweeps would raise to the 5000th power.  Three fused multiply-adds on the two sums; -c bC sits in
// the ring (st_arrive).  k.cx = ah, k.rx = bh, k.rb = e, k.beta = c.
template <int DV = 0>
__device__ __forceinline__ v2f st_value2s(v2f we, v2f sn, v2f b, const jacobi_row &k)
{
    if (DV == 3) return __builtin_elementwise_fma((v2f)k.cx, we, __builtin_elementwise_fma((v2f)k.rx, sn, __builtin_elementwise_fma((v2f)k.rb, sn, b)));
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
    unsigned fc[2][2];              // the class byte of the row in flight (looked up when its loads are issued: scalar)
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
        if (DV == 3) {
            // (the ring holds -c bC, +0.0 on land: L * w + 0 is +0.0 on a land centre whatever the sign of w, and w on water)
            const v2f w0 = st_value2s<3>(we0, sn0, bb.xy, k), w1 = st_value2s<3>(we1, sn1, bb.zw, k);
            asm("v_pk_fma_f32 %0, %1, %2, 0" : "=v"(o0) : "v"(L0), "v"(w0));
            asm("v_pk_fma_f32 %0, %1, %2, 0" : "=v"(o1) : "v"(L1), "v"(w1));
            return v4f{ o0.x, o0.y, o1.x, o1.y };
        }
        const v2f t0 = st_div2<DV>(we0, k.cx, k.rx) + st_div2<(DV != 0)>(sn0, k.cy, k.ry), t1 = st_div2<DV>(we1, k.cx, k.rx) + st_div2<(DV != 0)>(sn1, k.cy, k.ry);
        o0 = st_div2<DV>(__builtin_elementwise_fma(L0, t0, -bb.xy), k.beta, k.rb);
        o1 = st_div2<DV>(__builtin_elementwise_fma(L1, t1, -bb.zw), k.beta, k.rb);
    }
    return v4f{ o0.x, o0.y, o1.x, o1.y };
}

// one level: sweep S of rows yA = e0 + 2t - S and yB = yA + 1 from level S-1's rows yA-1 .. yB+1 (the pair made one step ago
// and the pair made in this step); K = t mod 2
// MODE 4, 5, 6 (the deep form's front wave, ocean_jacobi_deep.hip): as 1 / 0 / 2, but level 8's rows go, in pair order and whatever they
// are, into the LDS slots `xout` for the back wave instead of to memory
// MODE 0: the pair's rows as their classes say (tests and branches); 1: both open water; 2: both through the coast formula;
// 3: both open water and known to be rows the unit writes (st_open_unit): level 8 stores them through a scalar base and the
// lane's byte offsets sA, sB, without a range test
template <int S, int K, int MODE = 0, int DV = 0>
__device__ __forceinline__ void st_level(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const v4f *ktab,
                                         int lane, int t, int yA, int r0, int r1, bool lane_ok, int xcol,
                                         v4f (&bq)[9], v4f (&kq)[9], int kb = 0, const char *sbase = nullptr, unsigned sA = 0, unsigned sB = 0,
                                         v4f (*xout)[64] = nullptr)
{
    // bq[k], kq[k]: right-hand side and coefficients of ring row (newest - k); this level's rows are k = S and k = S - 1
    const v4f bA = bq[S], bB = bq[S - 1], kA = kq[S], kB = kq[S - 1];
    // the rows the next levels need are read from the rings while this level computes.  Level S + 1 works one row further
    // south: its second row is this level's first (already in registers), only its first row, k = S + 1, is new.
    if (S < 8) {
        // coefficients of row k = S + 1, window row 2t - (S + 1); every lane reads the same 16 bytes: one broadcast.
        // Unit form: kb = byte offset of window row 2t - 8 in a VGPR, set once per step -- the level adds a constant (in the
        // first four steps of a unit the offset is negative: the read lands in the ring in front of the table and is not used)
        kq[S + 1] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(ktab) + kb + (8 - (S + 1)) * 16);
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
    if (MODE == 2 || MODE == 6) {
        const int sa = st_wrap(z.slot - S), sb = sa + 1 == ST_RING ? 0 : sa + 1;
        oA = st_row_any<DV>(N0, C0, S0, bA, kA, z.cy, z.ry, true, cring[sa], lane);
        oB = st_row_any<DV>(N1, N0, C0, bB, kB, z.cy, z.ry, true, cring[sb], lane);
    } else if (MODE == 1 || MODE == 3 || MODE == 4 || !(((z.landbits | z.mixbits | (DV == 1 ? z.badbits : 0u)) >> S) & 3u)) {   // (one two-bit field of the step's combined mask: the common case costs one scalar test)
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
        if (DV == 1 && ((z.badbits >> S) & 3u)) {
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
    else if (MODE == 4 || MODE == 5 || MODE == 6) { xout[0][lane] = oA; xout[1][lane] = oB; }
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
__device__ __forceinline__ void st_fetch(const st_args &a, st_state &z, int set, int j, int e, int col, int colp, int tx = 0, unsigned cls = 2u)
{
    // Three scalar row bases and one 32-bit byte offset per lane: the loads take the SGPR-base + VGPR-offset form, whichever
    // way the bases were found -- advanced by a row (plain units: rows are fetched in order e0, e0 + 1, ...; a step spends a
    // third of its instructions outside its levels, most of them scalar address arithmetic: per-step clocks)
    // or mapped (mirror beyond a pole or an edge, clamp beyond the stored band).  One set of load instructions for both:
    // written as two branches that each load, the compiler merges them and builds every address in a VGPR pair.
    const char *bp, *bb, *bc;
    bool shift = false;
    if (z.plain && !ST_DBG(a, 1)) {
        bp = reinterpret_cast<const char *>(z.nx_p); bb = reinterpret_cast<const char *>(z.nx_b); bc = reinterpret_cast<const char *>(z.nx_c);
        z.nx_p += a.rs; z.nx_b += a.rs; z.nx_c += a.rs;
    } else {
        const st_row m = st_map_row(a, ST_DBG(a, 1) ? (e & 63) : e);
        const long rowoff = (long)m.l * a.rs;
        bp = reinterpret_cast<const char *>(a.pin + rowoff); bb = reinterpret_cast<const char *>(a.b + rowoff); bc = reinterpret_cast<const char *>(a.code + rowoff);
        shift = m.shift;
    }
    // an open-water row's codes are never looked at (st_arrive): its load goes to the band's first row, which every wave
    // keeps in the L2 -- the same instruction, a ninth of the row's bytes less from HBM
    if ((cls & 3u) == 0u) bc = reinterpret_cast<const char *>(a.code);
    const unsigned cc = (unsigned)(shift ? colp : col);
    z.ps[set][j] = *reinterpret_cast<const v4f *>(bp + cc * 4u);
    z.bs[set][j] = *reinterpret_cast<const v4f *>(bb + cc * 4u);
    z.cs[set][j] = *reinterpret_cast<const unsigned *>(bc + cc);
}
typedef const float __attribute__((address_space(4))) *st_kptr;

// the row that has arrived in staging slot [K][j]: classify it, take its pressure into level 0, park its right-hand side
// and codes in the rings
// DV = 3 (tolerance rendering): the right-hand side is parked as -c bC, c = the row's 1 / Beta (negc = -c), +0.0 on land
template <int K, int DV = 0>
__device__ __forceinline__ void st_arrive(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], int lane,
                                          int j, int e, unsigned cls, int slot, unsigned &land, unsigned &mix, float negc = 0.0f)
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
            if (DV == 3) { bq.xy = bq.xy * negc; bq.zw = bq.zw * negc; }
            const unsigned landb = DV == 3 ? 0u : 0x80000000u;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.x) : "v"(code_mask<4>(c, 0)), "v"(landb), "v"(bq.x));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.y) : "v"(code_mask<12>(c, 0)), "v"(landb), "v"(bq.y));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.z) : "v"(code_mask<20>(c, 0)), "v"(landb), "v"(bq.z));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bq.w) : "v"(code_mask<28>(c, 0)), "v"(landb), "v"(bq.w));
            st_park_pairs(&bring[slot][lane], bq);
            return;
        }
    }
    // (an open row among coast rows goes through the coast formula with the rest of its step: counts of zero)
    cring[slot][lane] = 0u;                                   // (a land row's entry is never read: no test)
    if (DV == 3) { v4f bq = z.bs[K][j]; bq.xy = bq.xy * negc; bq.zw = bq.zw * negc; st_park_pairs(&bring[slot][lane], bq); }
    else st_park_pairs(&bring[slot][lane], z.bs[K][j]);
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
        st_level<S, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - S, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (S < 8) {
            if (S == 2 * t) st_twin_hand_over<S & 7, K>(z, xme, xpeer, lane, false);
            else if (S == 2 * t + 1) st_twin_hand_over<S & 7, K>(z, xme, xpeer, lane, true);
        }
    }
}

template <int K, int DV = 0, int TW = 0>
__device__ __forceinline__ void st_step(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const v4f *ktab, int lane, int t,
                                        int e0, int nsteps, int r1, unsigned cls_lo, unsigned cls_hi, int col, int colp, bool lane_ok, int xcol, int tx = 0,
                                        v4f (*xme)[64] = nullptr, const v4f (*xpeer)[64] = nullptr)
{
    const int e_t = e0 + 2 * t;                              // first row of the pair that has just arrived
    const int slotB = z.slot + 1;                            // (ST_RING is even: a pair never wraps)
    unsigned landA, mixA, landB, mixB;
    const unsigned clsA = z.fc[K][0], clsB = z.fc[K][1];
#ifdef ST_EXPERIMENTS
    // DMG_ST_STEPPROF: slots 22..42 = the clock when the even step's rows have arrived (its wait for the staged loads is over),
    // slots 43..62 = when the even step's levels are done, of a unit's first 20 iterations (slots 0..21: top of the iteration)
    if (K == 0 && a.stepprof && lane == 0 && t < 40) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // (the even step's six loads are the oldest of the twelve in flight)
        a.stepprof[(size_t)(TW ? 2 * blockIdx.x + (threadIdx.x >> 6) : blockIdx.x) * 64 + 22 + t / 2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // (twin units: the partner has read the last row handed over -- ring slots 10 and 11 are the unit's own from here on)
    if (TW && K == 0 && t == TB_T / 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (tolerance rendering: the two rows' -1 / Beta from the table, window rows 2t and 2t + 1; every lane reads the same word)
    const float ncA = DV == 3 ? -reinterpret_cast<const float *>(&ktab[2 * t])[3] : 0.0f;
    const float ncB = DV == 3 ? -reinterpret_cast<const float *>(&ktab[2 * t + 1])[3] : 0.0f;
    st_arrive<K, DV>(a, z, bring, cring, lane, 0, e_t, clsA, z.slot, landA, mixA, ncA);
    st_arrive<K, DV>(a, z, bring, cring, lane, 1, e_t + 1, clsB, slotB, landB, mixB, ncB);
    if (TW && K == 0 && t == 0) st_twin_hand_over<0, K>(z, xme, xpeer, lane, false);      // "level 0": the partner's first row is this unit's row -1
    z.landbits = (z.landbits << 2) | (landA << 1) | landB;
    z.mixbits = (z.mixbits << 2) | (mixA << 1) | mixB;
    if (DV == 1) z.badbits = (z.badbits << 2) | ((clsA >> 1) & 2u) | ((clsB >> 2) & 1u);
    const unsigned special = DV == 1 ? z.badbits : 0u;           // rows that need the tested path like an all-land row does
    // the staging set is free again: the pair of step t + 2 (past the window's end the rows are clamped and unused)
    z.fc[K][0] = st_class4(cls_lo, 2 * t + 4); z.fc[K][1] = st_class4(cls_lo, 2 * t + 5);
    st_fetch(a, z, K, 0, e_t + 4, col, colp, tx, z.fc[K][0]);
    st_fetch(a, z, K, 1, e_t + 5, col, colp, tx, z.fc[K][1]);
    // level 1: rows e_t - 1 (in the ring since the last step) and e_t (just written)
    // (a twin unit's level 1 computes window row 0 in step 0: its table index is not clamped -- row -1's entry is the 16 bytes in
    // front of the table, read and not used)
    const int s1 = st_wrap(z.slot - 1), r1w = TW ? 2 * t - 1 : max(2 * t - 1, 0);
    v4f bq[9], kq[9];
    bq[1] = bring[s1][lane]; bq[0] = bring[z.slot][lane]; kq[1] = ktab[r1w]; kq[0] = ktab[r1w + 1];
    int kb = 0;
    asm volatile("v_mov_b32 %0, %1" : "=v"(kb) : "s"((2 * t - 8) * 16));
    // no land and no coast in any of the ten rows in flight (every step of a strip in open water): the eight levels without
    // a test or a branch between them -- one block for the scheduler; a branch per level visit costs the kernel ~3 %
    // A unit's first eight steps fill the pipeline: level S has its first rows to compute at step S.  Those steps take the
    // tested path whatever their rows are, and there a level that has nothing to do yet is skipped outright (round 3: per-step
    // clocks showed a skipped level visit costing 0.25 us through the all-land path -- as much as a real one -- and a unit's
    // eight fill steps 2.1 us each; with the skip a unit is ~7 us shorter)
    const bool filled = t >= (TW ? TB_T / 2 : TB_T);
    const int r0 = TW ? e0 : e0 + TB_T;                      // the unit's first output row
    if (filled && !((z.landbits | z.mixbits | special) & 0x3FFu)) {
        st_level<1, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<2, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<3, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<4, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<5, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<6, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<7, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<8, K, 1, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    } else if (filled && !((z.landbits | special) & 0x3FFu)) {
        // coast among the ten rows, but no row that is land everywhere (coast rows come in long runs: 18 % of the survey
        // terrain's steps): every row through the coast formula, open rows with counts of zero -- again no test, no branch
        st_level<1, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<2, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<3, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<4, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<5, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<6, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<7, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        st_level<8, K, 2, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
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
        if (1 <= t) st_level<1, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (2 <= t) st_level<2, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (3 <= t) st_level<3, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (4 <= t) st_level<4, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (5 <= t) st_level<5, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (6 <= t) st_level<6, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (7 <= t) st_level<7, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
        if (8 <= t) st_level<8, K, 0, DV>(a, z, bring, cring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb);
    }
    z.slot = z.slot + 2 == ST_RING ? 0 : z.slot + 2;
#ifdef ST_EXPERIMENTS
    if (K == 0 && a.stepprof && lane == 0 && t < 40) a.stepprof[(size_t)(TW ? 2 * blockIdx.x + (threadIdx.x >> 6) : blockIdx.x) * 64 + 43 + t / 2] = __builtin_amdgcn_s_memrealtime();
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
        if (DV == 3) {
            const v4f kc = ktab[2 * t + j];
            const float negc = -kc.w;
            v4f bq = z.bs[K][j]; bq.xy = bq.xy * negc; bq.zw = bq.zw * negc;
            st_park_pairs(&bring[z.slot + j][lane], bq);
        }
        else st_park_pairs(&bring[z.slot + j][lane], z.bs[K][j]);
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
    if (PHASE != 0 || 1 <= t) st_level<1, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 1, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 2 <= t) st_level<2, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 2, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 3 <= t) st_level<3, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 3, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 4 <= t) st_level<4, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 4, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 5 <= t) st_level<5, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 5, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 6 <= t) st_level<6, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 6, r0, r1, lane_ok, xcol, bq, kq, kb);
    if (PHASE != 0 || 7 <= t) st_level<7, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 7, r0, r1, lane_ok, xcol, bq, kq, kb);
    // (voA, voB address rows e_t + 4, e_t + 5; level 8's rows are e_t - 8, e_t - 7: sb = pout - 12 rows)
    if (PHASE != 0) st_level<8, K, M, DV>(a, z, bring, nocring, ktab, lane, t, e_t - 8, r0, r1, lane_ok, xcol, bq, kq, kb, sb, voA, voB);
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

// A unit's tables: the classes of its window rows (four per lane: st_class4), the flags of rows without an exact divisor pair, the
// coefficient table in LDS.  Round 6: every load unconditional (addresses clamped, the choice made afterwards) and all twelve in
// flight together behind ONE wait.  Written as `if (inside) cls = table[...]` the class loads came out as four conditional loads
// with an s_waitcnt vmcnt(0) after each -- four memory round trips in a row, the first of them behind the unit's first row
// requests, then a fifth for the flags and a sixth for the coefficients: 5.5 us of a 45-us unit of a one-round launch
// (profiles/r06_band_units.txt, per-step clocks).
template <int DV>
__device__ __forceinline__ void st_unit_tables(const st_args &a, int lane, int tx, int e0, int n, unsigned &cls_lo, v4f *ktab)
{
    unsigned cb[4], bd[4];
    v4f kt[4];
    bool inside[4];
    const unsigned char *rc = a.rowclass ? a.rowclass + (size_t)tx * a.L : a.code;      // (no class table: any readable byte, not used)
    const unsigned char *rb = a.rowbad ? a.rowbad : a.code;                             // (likewise)
    const unsigned badmask = (DV == 1 && a.rowbad) ? 1u : 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int lu = e0 + lane + 64 * q, yg = a.y_first + lu;
        inside[q] = yg >= 0 && yg < a.H && lu >= 0 && lu < a.L;
        cb[q] = rc[inside[q] ? lu : 0];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) bd[q] = DV == 1 ? (unsigned)rb[st_map_row(a, e0 + lane + 64 * q).l] : 0u;     // (mirrored and clamped rows have their image's divisors)
    // one 16-byte table entry per window row (two spare: the next level's prefetch)
#pragma unroll
    for (int q = 0; q < 4; q++) kt[q] = st_kswap(*reinterpret_cast<const v4f *>(a.jcoef + (size_t)st_map_row(a, e0 + min(lane + 64 * q, n - 1)).l * 8));
    // classes of the window rows: rows beyond the grid or the stored band count as mixed (their codes decide)
    cls_lo = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) cls_lo |= (((a.rowclass && inside[q]) ? cb[q] : 2u) | ((bd[q] & badmask) << 2)) << (8 * q);
#pragma unroll
    for (int q = 0; q < 4; q++) if (lane + 64 * q < n) ktab[lane + 64 * q] = kt[q];
}

// constants of the kernel forms that the host-side planner (ocean_jacobi_stream_plan.hip) needs as well
#define ST_TWIN_MAXROWS 232
