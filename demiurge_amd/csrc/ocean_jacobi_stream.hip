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
//     step t:  row e_t = r0 - 8 + t arrives from HBM (level 0)
//              level s (1..8) computes sweep s of row e_t - s from the three level s-1 rows e_t-s-1, e_t-s, e_t-s+1,
//              the last of which level s-1 has produced in this very step
//              level 8's row e_t - 8 is final and is stored
//
// A wave keeps, per level, the two previous rows in registers (a ring of three slots, the loop is unrolled by three so
// that every slot is a fixed register), the right-hand side and the land codes of the 9 rows in flight in a private
// LDS ring, and the per-row coefficients in SGPRs.  East/west neighbours are adjacent lanes (DPP), north/south are the
// wave's own registers: no LDS exchange, no barrier, every wave has the same work, loads and stores trickle along with
// the arithmetic.  A work unit is a strip segment of up to 112 rows: rows [r0-8, r1+8) are read, and level s is only
// run where its rows are still needed ([r0-8+s, r1+8-s)), so the recomputation is 7/n of the work (n = rows per unit)
// instead of 16/64.  Units whose rows are land everywhere are not dispatched; land rows inside a unit cost nothing.
//
// Canvas topology exactly as in ocean_jacobi_tb.hip: columns wrap, rows beyond a pole are the same rows mirrored and
// shifted by W/2 (offset(), src/Shader.h:86-95), rows beyond a non-pole edge are plain mirrors with the clamp bit
// cleared.  Arithmetic: jacobi_value() of ocean_internal.h, the same instruction sequence as every other Jacobi kernel.
#include "ocean_internal.h"
#include "ocean_tb.h"

#define ST_MAXROWS 112            // output rows per unit: the 128 window rows' classes fit two registers
#define ST_RING 9                 // rows in flight between level 0 and level 8

struct st_args {
    const float *pin;
    float *pout;
    const float *b;
    const unsigned char *code;
    const float *jcoef;             // [L][8] = cx, rx, beta, rb, cy, ry, -, -  per local row (r* = correctly rounded reciprocals)
    int W, L;
    int y_first, H;                 // grid row of local row 0; grid height
    int spole, npole;
    const unsigned char *rowclass;  // [tiles_x][L]: 0 open ocean, 1 land everywhere, 2 mixed; or null
    const int4 *units;              // {tile column, first output row, end output row, -}
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
__device__ __forceinline__ v2f st_div2(v2f x, float c, float r)
{
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
    v4f w[8][3];                   // w[s][k]: rows of level s (after s sweeps), ring of three
    v4f bnext;                      // right-hand side and (raw) codes of the row that arrives at the next step
    unsigned cnext;
    unsigned landbits, mixbits;     // bit j: row e_t - j is land everywhere / has coast
    int slot;                       // t mod ST_RING
    v4f knext;                      // its coefficients {cx, 1/cx, Beta, 1/Beta}
    float cy, ry;                   // the same for every row (pixelsize().y, src/Shader.h:71-74): two SGPRs
};

// one level: sweep S of row (e_t - S) from level S-1's rows; K = t mod 3
template <int S, int K>
__device__ __forceinline__ void st_level(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], const v4f *kring,
                                         int lane, int t, int e_t, bool lane_ok, int xcol, const v4f bb, v4f &bb_next, const v4f kk, v4f &kk_next)
{
    // the next level's right-hand side and coefficients are read from the rings while this level computes
    if (S < 8) {
        int sn = z.slot - (S + 1);
        if (sn < 0) sn += ST_RING;
        bb_next = bring[sn][lane];
        kk_next = kring[sn];                                 // every lane reads the same 16 bytes: one broadcast
    }
    if (t < 2 * S) return;                                   // level S starts once its first needed row exists
    if ((z.landbits >> S) & 1u) {                            // land everywhere: exactly 0, neither computed nor stored
        if (S < 8) z.w[S & 7][K] = (v4f)(0.0f);
        return;
    }
    const jacobi_row k = { kk.x, z.cy, kk.z, kk.y, z.ry, kk.w };
    const v4f Sv = z.w[S - 1][(K + 1) % 3], C = z.w[S - 1][(K + 2) % 3], N = z.w[S - 1][K];
    const float wl = dpp_from_prev_lane(C.w);                // west neighbour of C.x
    const float er = dpp_from_next_lane(C.x);                // east neighbour of C.w
    v2f o0, o1;
    if (!((z.mixbits >> S) & 1u)) {                          // wave-uniform: open ocean, no selects
        const v2f Wlo = { wl, C.x }, Mid = { C.y, C.z }, Ehi = { C.w, er };
        o0 = st_value2(N.xy, Sv.xy, Mid, Wlo, bb.xy, k);     // cells x, y: W = (wl, x), E = (y, z)
        o1 = st_value2(N.zw, Sv.zw, Ehi, Mid, bb.zw, k);     // cells z, w: W = (y, z), E = (w, er)
    } else {
        int sl = z.slot - S;
        if (sl < 0) sl += ST_RING;
        const unsigned c = cring[sl][lane];
        const v2f n0 = { st_sel<0>(c, t, C.x, N.x), st_sel<8>(c, t, C.y, N.y) }, n1 = { st_sel<16>(c, t, C.z, N.z), st_sel<24>(c, t, C.w, N.w) };
        const v2f s0 = { st_sel<1>(c, t, C.x, Sv.x), st_sel<9>(c, t, C.y, Sv.y) }, s1 = { st_sel<17>(c, t, C.z, Sv.z), st_sel<25>(c, t, C.w, Sv.w) };
        const v2f e0 = { st_sel<2>(c, t, C.x, C.y), st_sel<10>(c, t, C.y, C.z) }, e1 = { st_sel<18>(c, t, C.z, C.w), st_sel<26>(c, t, C.w, er) };
        const v2f w0 = { st_sel<3>(c, t, C.x, wl), st_sel<11>(c, t, C.y, C.x) }, w1 = { st_sel<19>(c, t, C.z, C.y), st_sel<27>(c, t, C.w, C.z) };
        const v2f r0 = st_value2(n0, s0, e0, w0, bb.xy, k), r1 = st_value2(n1, s1, e1, w1, bb.zw, k);
        o0 = v2f{ st_zero_land<4>(c, t, r0.x), st_zero_land<12>(c, t, r0.y) };
        o1 = v2f{ st_zero_land<20>(c, t, r1.x), st_zero_land<28>(c, t, r1.y) };
    }
    const v4f o = { o0.x, o0.y, o1.x, o1.y };
    if (S < 8) z.w[S & 7][K] = o;
    else if (lane_ok)
        __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(a.pout + (size_t)(e_t - 8) * a.W + xcol));   // never re-read in this launch
}

// class of window row i (0..127) from the two class registers (lane i holds rows i and 64 + i)
__device__ __forceinline__ unsigned st_class(unsigned cls_lo, unsigned cls_hi, int i)
{
    return (unsigned)__builtin_amdgcn_readlane((int)(i < 64 ? cls_lo : cls_hi), i & 63);
}

// issue the loads of window row e (arrives at the next step): p into `dst`, b and codes into the state
__device__ __forceinline__ void st_fetch(const st_args &a, st_state &z, v4f &dst, int e, unsigned cls, int col, int colp)
{
    const st_row m = st_map_row(a, e);
    if (cls == 1u) return;                                   // land everywhere: nothing to read (its coefficients are never used)
    const size_t off = (size_t)m.l * a.W + (m.shift ? colp : col);
    dst = *reinterpret_cast<const v4f *>(a.pin + off);
    z.bnext = *reinterpret_cast<const v4f *>(a.b + off);
    z.knext = *reinterpret_cast<const v4f *>(a.jcoef + (size_t)m.l * 8);            // one address for the whole wave
    if (cls == 2u) z.cnext = *reinterpret_cast<const unsigned *>(a.code + off);     // looked at when the row has arrived
}
typedef const float __attribute__((address_space(4))) *st_kptr;

template <int K>
__device__ __forceinline__ void st_step(const st_args &a, st_state &z, v4f (*bring)[64], unsigned (*cring)[64], v4f *kring, int lane, int t,
                                        int e0, int nsteps, unsigned cls_lo, unsigned cls_hi, int col, int colp, bool lane_ok, int xcol)
{
    const int e_t = e0 + t;
    // level 1's right-hand side (row e_t - 1) is in the ring already
    int s1 = z.slot - 1;
    if (s1 < 0) s1 += ST_RING;
    v4f bcur = bring[s1][lane], bnxt, kcur = kring[s1], knxt;
    // the row that has just arrived: classify it, park its right-hand side and codes in the ring
    const unsigned cls = st_class(cls_lo, cls_hi, t);
    unsigned land = cls == 1u, mix = 0;
    if (cls == 1u) z.w[0][K] = (v4f)(0.0f);
    else if (cls == 2u) {
        unsigned c = z.cnext;
        const st_row m = st_map_row(a, e_t);
        if (m.flip) c = (c & 0xFCFCFCFCu) | ((c & 0x01010101u) << 1) | ((c & 0x02020202u) >> 1);   // mirrored row: swap N/S bits
        // a non-pole edge row's clamp bit only says "the tap lands on the cell itself"; its mirror row is in the window
        const int yg = a.y_first + e_t;
        if (yg == 0 && !a.spole) c &= ~0x02020202u;
        if (yg == a.H - 1 && !a.npole) c &= ~0x01010101u;
        mix = __builtin_amdgcn_ballot_w64((c & 0x1F1F1F1Fu) != 0) != 0;
        land = __builtin_amdgcn_ballot_w64((c & 0x10101010u) != 0x10101010u) == 0;
        if (mix && !land) cring[z.slot][lane] = c;
    }
    if (!land) {
        bring[z.slot][lane] = z.bnext;
        if (lane == 0) kring[z.slot] = z.knext;
    }
    z.landbits = (z.landbits << 1) | land;
    z.mixbits = (z.mixbits << 1) | (mix & ~land);
    st_level<1, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bcur, bnxt, kcur, knxt);
    // level 1 was the last reader of w[0][(K+1)%3]: the next row lands there
    if (t + 1 < nsteps) st_fetch(a, z, z.w[0][(K + 1) % 3], e_t + 1, st_class(cls_lo, cls_hi, t + 1), col, colp);
    st_level<2, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bnxt, bcur, knxt, kcur);
    st_level<3, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bcur, bnxt, kcur, knxt);
    st_level<4, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bnxt, bcur, knxt, kcur);
    st_level<5, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bcur, bnxt, kcur, knxt);
    st_level<6, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bnxt, bcur, knxt, kcur);
    st_level<7, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bcur, bnxt, kcur, knxt);
    st_level<8, K>(a, z, bring, cring, kring, lane, t, e_t, lane_ok, xcol, bnxt, bcur, knxt, kcur);
    z.slot = z.slot == ST_RING - 1 ? 0 : z.slot + 1;
}

// one wave per unit; 3 waves per SIMD leave 168 registers each
__global__ __launch_bounds__(64, 3) void k_jacobi_stream(st_args a)
{
    __shared__ v4f bring[ST_RING][64];
    __shared__ unsigned cring[ST_RING][64];
    __shared__ v4f kring[ST_RING];
    const int lane = threadIdx.x;
    const int4 u = a.units[blockIdx.x];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const int W = a.W;
    const int nsteps = (r1 - r0) + 2 * TB_T;
    const int e0 = r0 - TB_T;                       // unclamped local row of the first window row
    int col = tx * TB_OW - TB_T + 4 * lane;         // this lane's 4 columns (wrapped), plain and pole-shifted
    const int xcol = col;                           // unwrapped, for the store
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);

    // classes of the window rows: rows beyond the grid or the stored band count as mixed (their codes decide)
    unsigned cls_lo = 2u, cls_hi = 2u;
    if (a.rowclass) {
        int lu = e0 + lane, yg = a.y_first + lu;
        if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls_lo = a.rowclass[(size_t)tx * a.L + lu];
        lu += 64; yg += 64;
        if (lane + 64 < nsteps && yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls_hi = a.rowclass[(size_t)tx * a.L + lu];
    }

    st_state z;
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 3; k++) z.w[s][k] = (v4f)(0.0f);
    z.bnext = (v4f)(0.0f);
    z.cnext = 0;
    z.landbits = 0; z.mixbits = 0; z.slot = 0;
    z.knext = (v4f)(1.0f);
    {
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    st_fetch(a, z, z.w[0][0], e0, st_class(cls_lo, cls_hi, 0), col, colp);
    for (int t = 0; t < nsteps; t += 3) {
        st_step<0>(a, z, bring, cring, kring, lane, t, e0, nsteps, cls_lo, cls_hi, col, colp, lane_ok, xcol);
        if (t + 1 < nsteps) st_step<1>(a, z, bring, cring, kring, lane, t + 1, e0, nsteps, cls_lo, cls_hi, col, colp, lane_ok, xcol);
        if (t + 2 < nsteps) st_step<2>(a, z, bring, cring, kring, lane, t + 2, e0, nsteps, cls_lo, cls_hi, col, colp, lane_ok, xcol);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------
bool ocean_stream_enabled()
{
    static const bool on = [] { const char *e = getenv("DMG_JACOBI_STREAM"); return !e || atoi(e) != 0; }();
    return on;
}

static int st_unit_rows()
{
    static const int rows = [] {
        const char *e = getenv("DMG_ST_ROWS");
        int v = e ? atoi(e) : 96;
        return v >= 8 && v <= ST_MAXROWS ? v : 96;
    }();
    return rows;
}

// Units of a launch over local rows [lo, hi): every strip cut into segments of st_unit_rows() rows, each trimmed to its
// first / last row that is not land everywhere; empty ones are dropped.  part 1 / 2: only the units that do / do not
// produce rows of [lo, zone_lo) or [zone_hi, hi) (the two halves of a launch split around a halo exchange).
static const dmg_ocean::tile_list *st_units(dmg_ocean *o, int lo, int hi, int part, int zone_lo, int zone_hi, int seg)
{
    auto key = std::make_tuple(lo, hi, -seg, part, zone_lo, zone_hi);    // "rows per wave" < 0 = stream units of -that many rows
    auto it = o->tb_lists.find(key);
    if (it != o->tb_lists.end()) return &it->second;
    const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW, L = o->L;
    const bool classes = !o->rowclass_host.empty();
    std::vector<int> units;                              // int4 per unit
    units.reserve((size_t)tiles_x * ((hi - lo) / seg + 1) * 4);
    for (int s0 = lo; s0 < hi; s0 += seg) {
        const int s1 = std::min(s0 + seg, hi);
        for (int tx = 0; tx < tiles_x; tx++) {
            int r0 = s0, r1 = s1;
            if (classes) {
                const unsigned char *cls = o->rowclass_host.data() + (size_t)tx * L;
                while (r0 < r1 && cls[r0] == 1) r0++;
                while (r1 > r0 && cls[r1 - 1] == 1) r1--;
            }
            if (r0 >= r1) continue;
            if (part) {
                const bool edge = r0 < zone_lo || r1 > zone_hi;
                if (edge != (part == 1)) continue;
            }
            units.push_back(tx); units.push_back(r0); units.push_back(r1); units.push_back(0);
        }
    }
    dmg_ocean::tile_list tl;
    tl.n = (int)(units.size() / 4);
    if (tl.n && hipMalloc((void **)&tl.d, units.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (tl.n && hipMemcpy(tl.d, units.data(), units.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return &(o->tb_lists[key] = tl);
}

// 8 sweeps over local rows [lo, hi) from p[from] into p[from ^ 1]; seg = rows per unit (0: the default)
int ocean_launch_stream(dmg_ocean *o, int lo, int hi, int part, int zone_lo, int zone_hi, int from, hipStream_t st, int seg)
{
    if (seg <= 0) seg = st_unit_rows();
    if (seg > ST_MAXROWS) seg = ST_MAXROWS;
    const dmg_ocean::tile_list *tl = st_units(o, lo, hi, part, zone_lo, zone_hi, seg);
    if (!tl) return dmg_fail(DMG_ENOMEM, "unit list allocation failed");
    if (tl->n == 0) return DMG_OK;
    st_args a;
    a.pin = o->p[from];
    a.pout = o->p[from ^ 1];
    a.b = o->divw;
    a.code = o->code;
    a.jcoef = o->jcoef;
    a.W = o->grid.W; a.L = o->L;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    static const bool use_cls = getenv("DMG_TB_NOCLASS") == nullptr;
    a.rowclass = use_cls ? o->rowclass : nullptr;
    a.units = (const int4 *)tl->d;
    hipLaunchKernelGGL(k_jacobi_stream, dim3((unsigned)tl->n), dim3(64), 0, st, a);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}
