// Temporally blocked velocity diffusion: up to TB_T = 8 sweeps of the diffusion shader (reference
// src/filter/OceanCurrents.cpp:664-693, the 50-sweep loop :701-707) per launch, bit-identical to 8 launches of
// k_diffusion (ocean_passes.hip).
//
// Same scheme as the blocked Jacobi kernel (ocean_jacobi_tb.hip): a workgroup loads a 256-column x (R*NW)-row tile
// with an 8-cell halo once, advances it 8 sweeps in registers (east/west neighbours through DPP, the rows between
// waves through LDS) and writes the interior once: ~21 B of HBM traffic per cell per 8 sweeps instead of 8 x 17.
// What differs from the pressure solve:
//   * two components per cell, (vx, vy) interleaved in memory like the reference's RG32F texture; a thread keeps its
//     4 x R cells as separate x and y registers, de-interleaved at load;
//   * a tap across a pole is sign-flipped (get_velocity(), :657): rows beyond a pole enter the tile mirrored, shifted
//     by W/2 AND negated -- the stencil is linear without a constant term, so the images stay exact negated images;
//   * land cells are computed like any other cell (no `solid -> 0` in this shader) and hold arbitrary values, so the
//     "land neighbour -> centre value" rule (:682-685) is a true select (v_bfi_b32) and no row can be skipped;
//   * the right-hand side is the cell's own vx for BOTH components (`float bC = texture(v,st).r`, :687; quirk flag).
// The division by the row constant Beta uses the row's correctly rounded reciprocal with Markstein's correction
// (dmg_div_rc), which is the IEEE quotient unless the remainder underflows (0 < |numerator| < 2^-100).  Land cells
// next to a coast pick up whatever the neighbouring water holds, down to denormals when the host seeds such a field,
// so every wave-row that has land in or next to it checks its eight numerators (min over |n|, then a closer look only
// if something is below the threshold) and divides the row with the IEEE sequence when one of them is that small.
// Wave-rows without land skip the check: there a numerator in (0, 2^-100) -- six orders of magnitude below anything
// the wind term leaves in open water -- may round its quotient one ulp (1e-37) differently from the single-sweep kernel.
#include "ocean_internal.h"
#include "ocean_tb.h"

struct dtb_args {
    const float2 *vin;
    float2 *vout;
    const unsigned char *code;
    const float *px, *py, *beta, *rb;   // per local row
    int W, L;
    int out_lo, out_hi;                 // local rows written by this launch
    int y_first, H;                     // grid row of local row 0; grid height
    int spole, npole;
    int nsweeps;                        // 1..TB_T
    int tiles_x, tiles_y;
    const unsigned char *rowclass;      // [tiles_x][L]: 0 = no land in or next to the wave-row (codes all zero), or null
    int scalar_rhs;
};

struct dif_row { float px, py, beta, rb; };

// ((bC + (pW + pE)*px) + (pS + pN)*py) in the association Mesa gives :693 (see k_diffusion)
__device__ __forceinline__ float dif_numerator(float pN, float pS, float pE, float pW, float b, const dif_row &k)
{
    return __fadd_rn(__fadd_rn(b, __fmul_rn(__fadd_rn(pW, pE), k.px)), __fmul_rn(__fadd_rn(pS, pN), k.py));
}
// (mask & pC) | (~mask & pK): the centre value where the neighbour is land
__device__ __forceinline__ float pick_centre(int mask, float pC, float pK)
{
    float o;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(o) : "v"(mask), "v"(pC), "v"(pK));
    return o;
}
template <int SHIFT>
__device__ __forceinline__ void dif_numerators_coded(float Cx, float Cy, float Nx, float Ny, float Sx, float Sy, float Ex, float Ey,
                                                     float Wx, float Wy, float by, unsigned code, int tick, const dif_row &k,
                                                     float *nx, float *ny)
{
    const int mN = code_mask<SHIFT + 0>(code, tick), mS = code_mask<SHIFT + 1>(code, tick);
    const int mE = code_mask<SHIFT + 2>(code, tick), mW = code_mask<SHIFT + 3>(code, tick);
    *nx = dif_numerator(pick_centre(mN, Cx, Nx), pick_centre(mS, Cx, Sx), pick_centre(mE, Cx, Ex), pick_centre(mW, Cx, Wx), Cx, k);
    *ny = dif_numerator(pick_centre(mN, Cy, Ny), pick_centre(mS, Cy, Sy), pick_centre(mE, Cy, Ey), pick_centre(mW, Cy, Wy), by, k);
}

// maps an extended-tile row to its source: local row l, whether it is a mirrored copy, whether it lies beyond a pole
struct dtb_row { int l; bool flip, pole; };
__device__ __forceinline__ dtb_row dtb_map_row(const dtb_args &a, int local_unclamped)
{
    dtb_row m;
    int y = a.y_first + local_unclamped;
    m.flip = false; m.pole = false;
    if (y < 0) { y = -1 - y; m.flip = true; m.pole = a.spole != 0; }
    else if (y >= a.H) { y = 2 * a.H - 1 - y; m.flip = true; m.pole = a.npole != 0; }
    int l = y - a.y_first;
    m.l = min(max(l, 0), a.L - 1);      // rows past the stored band only feed cells that are not written
    return m;
}

template <int R, int NW>
__global__ __launch_bounds__(NW * 64, (R <= 8 ? 4 : 2)) void k_diffusion_tb(dtb_args a)
{
    constexpr int EH = R * NW;                      // extended tile rows
    constexpr int OH = EH - 2 * TB_T;               // output rows per tile
    __shared__ float4 halo[2][NW][2][2][64];        // [sweep parity][wave][0 = row r=0, 1 = row r=R-1][component][lane]
    __shared__ float4 coef[EH];                     // per extended row: {px, py, beta, 1/beta}

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = a.W;
    const int ty = (int)blockIdx.x / a.tiles_x, tx = (int)blockIdx.x - ty * a.tiles_x;
    const int X0 = tx * TB_OW;                      // first output column of the tile
    const int Y0 = a.out_lo + ty * OH;              // first output local row of the tile
    const int e0 = Y0 - TB_T + wave * R;            // unclamped local row of this wave's register row 0

    int col = X0 - TB_T + 4 * lane;                 // this lane's 4 columns (wrapped), plain and pole-shifted
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;

    if (threadIdx.x < EH) {                         // one thread per extended row fills the coefficient table
        dtb_row m = dtb_map_row(a, Y0 - TB_T + (int)threadIdx.x);
        coef[threadIdx.x] = make_float4(a.px[m.l], a.py[m.l], a.beta[m.l], a.rb[m.l]);
    }

    float4 vx[R], vy[R];
    unsigned code[R];
    unsigned need_code = (1u << R) - 1u;            // wave-rows whose codes are not known to be all zero
    if (a.rowclass) {
        int rr = min(lane, R - 1);
        int lu = e0 + rr, yg = a.y_first + lu;
        unsigned cls = 2;
        if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls = a.rowclass[(size_t)tx * a.L + lu];
        unsigned long long oc = __builtin_amdgcn_ballot_w64(lane < R && cls == 0);
        need_code &= ~(unsigned)__builtin_amdgcn_readfirstlane((unsigned)oc);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        dtb_row m = dtb_map_row(a, e0 + r);
        size_t off = (size_t)m.l * W + (m.pole ? colp : col);
        code[r] = 0;
        if (need_code & (1u << r)) {
            unsigned c = *reinterpret_cast<const unsigned *>(a.code + off);
            if (m.flip) c = (c & 0xFCFCFCFCu) | ((c & 0x01010101u) << 1) | ((c & 0x02020202u) >> 1);   // swap N/S bits
            code[r] = c;
        }
        const float4 *src = reinterpret_cast<const float4 *>(a.vin + off);      // 4 cells = 2 x float4, (x,y) interleaved
        float4 q0 = src[0], q1 = src[1];
        vx[r] = make_float4(q0.x, q0.z, q1.x, q1.z);
        vy[r] = make_float4(q0.y, q0.w, q1.y, q1.w);
        if (m.pole) {                                                            // get_velocity(): v = -v across a pole (:657)
            vx[r] = make_float4(-vx[r].x, -vx[r].y, -vx[r].z, -vx[r].w);
            vy[r] = make_float4(-vy[r].x, -vy[r].y, -vy[r].z, -vy[r].w);
        }
    }
    unsigned mixed_rows = 0;                        // bit r: some cell of the wave-row has a land (or clamped) neighbour
#pragma unroll
    for (int r = 0; r < R; r++)
        if ((need_code & (1u << r)) && __builtin_amdgcn_ballot_w64((code[r] & 0x0F0F0F0Fu) != 0)) mixed_rows |= (1u << r);
    mixed_rows = __builtin_amdgcn_readfirstlane(mixed_rows);
    const bool srhs = a.scalar_rhs != 0;

    for (int s = 0; s < a.nsweeps; s++) {
        // publish this wave's first and last rows (values before the sweep), fetch the neighbours' rows
        const int par = s & 1;
        halo[par][wave][0][0][lane] = vx[0];
        halo[par][wave][0][1][lane] = vy[0];
        halo[par][wave][1][0][lane] = vx[R - 1];
        halo[par][wave][1][1][lane] = vy[R - 1];
        __syncthreads();
        // the outermost waves read a slot of their own: those rows are halo, their values never reach the output
        const int ws = wave > 0 ? wave - 1 : 0, wn = wave < NW - 1 ? wave + 1 : NW - 1;
        float4 prevx = halo[par][ws][1][0][lane], prevy = halo[par][ws][1][1][lane];      // row below r = 0
        // (the row above r = R-1 is read where it is used, at the loop's last row: held from here on its eight registers were what
        // pushed the default shape, R = 7, over its 128 -- 16 bytes of scratch per lane through round 5.  The slot is not written
        // again before the barrier of the sweep after next.)

        // rows closer than s+1 to the top/bottom edge of the extended tile are stale after this sweep whatever we do
        unsigned skip = 0;
        if (wave == 0) skip |= (2u << s) - 1u;                                  // rows r <= s (s <= 7)
        if (wave == NW - 1) { const int keep = R - 1 - s; skip |= keep <= 0 ? ~0u : ~((1u << keep) - 1u); }   // rows r >= R-1-s

#pragma unroll
        for (int r = 0; r < R; r++) {
            const float4 Cx = vx[r], Cy = vy[r];
            const float4 Nx = (r < R - 1) ? vx[r + 1] : halo[par][wn][0][0][lane], Ny = (r < R - 1) ? vy[r + 1] : halo[par][wn][0][1][lane];
            const float4 Sx = prevx, Sy = prevy;
            prevx = Cx; prevy = Cy;
            if (skip & (1u << r)) continue;
            const float wlx = dpp_from_prev_lane(Cx.w), wly = dpp_from_prev_lane(Cy.w);    // west neighbour of C.x
            const float erx = dpp_from_next_lane(Cx.x), ery = dpp_from_next_lane(Cy.x);    // east neighbour of C.w
            const float4 kk = coef[wave * R + r];
            const dif_row k = { kk.x, kk.y, kk.z, kk.w };
            const float4 By = srhs ? Cx : Cy;                                   // :687
            float4 nx, ny;
            if (!(mixed_rows & (1u << r))) {        // wave-uniform: no land neighbour anywhere in the row, no selects
                nx.x = dif_numerator(Nx.x, Sx.x, Cx.y, wlx, Cx.x, k);
                nx.y = dif_numerator(Nx.y, Sx.y, Cx.z, Cx.x, Cx.y, k);
                nx.z = dif_numerator(Nx.z, Sx.z, Cx.w, Cx.y, Cx.z, k);
                nx.w = dif_numerator(Nx.w, Sx.w, erx, Cx.z, Cx.w, k);
                ny.x = dif_numerator(Ny.x, Sy.x, Cy.y, wly, By.x, k);
                ny.y = dif_numerator(Ny.y, Sy.y, Cy.z, Cy.x, By.y, k);
                ny.z = dif_numerator(Ny.z, Sy.z, Cy.w, Cy.y, By.z, k);
                ny.w = dif_numerator(Ny.w, Sy.w, ery, Cy.z, By.w, k);
            } else {
                const unsigned c = code[r];
                dif_numerators_coded<0>(Cx.x, Cy.x, Nx.x, Ny.x, Sx.x, Sy.x, Cx.y, Cy.y, wlx, wly, By.x, c, s, k, &nx.x, &ny.x);
                dif_numerators_coded<8>(Cx.y, Cy.y, Nx.y, Ny.y, Sx.y, Sy.y, Cx.z, Cy.z, Cx.x, Cy.x, By.y, c, s, k, &nx.y, &ny.y);
                dif_numerators_coded<16>(Cx.z, Cy.z, Nx.z, Ny.z, Sx.z, Sy.z, Cx.w, Cy.w, Cx.y, Cy.y, By.z, c, s, k, &nx.z, &ny.z);
                dif_numerators_coded<24>(Cx.w, Cy.w, Nx.w, Ny.w, Sx.w, Sy.w, erx, ery, Cx.z, Cy.z, By.w, c, s, k, &nx.w, &ny.w);
                // a numerator in (0, 2^-100) would lose bits of its remainder: IEEE division for this row then
                const float thr = 0x1p-100f;
                float m = fminf(fminf(fminf(fabsf(nx.x), fabsf(nx.y)), fminf(fabsf(nx.z), fabsf(nx.w))),
                                fminf(fminf(fabsf(ny.x), fabsf(ny.y)), fminf(fabsf(ny.z), fabsf(ny.w))));
                if (__builtin_amdgcn_ballot_w64(m < thr)) {                     // something is tiny or zero: look closer
                    bool tiny = (nx.x != 0.0f && fabsf(nx.x) < thr) || (nx.y != 0.0f && fabsf(nx.y) < thr) ||
                                (nx.z != 0.0f && fabsf(nx.z) < thr) || (nx.w != 0.0f && fabsf(nx.w) < thr) ||
                                (ny.x != 0.0f && fabsf(ny.x) < thr) || (ny.y != 0.0f && fabsf(ny.y) < thr) ||
                                (ny.z != 0.0f && fabsf(ny.z) < thr) || (ny.w != 0.0f && fabsf(ny.w) < thr);
                    if (__builtin_amdgcn_ballot_w64(tiny)) {
                        vx[r] = make_float4(__fdiv_rn(nx.x, k.beta), __fdiv_rn(nx.y, k.beta), __fdiv_rn(nx.z, k.beta), __fdiv_rn(nx.w, k.beta));
                        vy[r] = make_float4(__fdiv_rn(ny.x, k.beta), __fdiv_rn(ny.y, k.beta), __fdiv_rn(ny.z, k.beta), __fdiv_rn(ny.w, k.beta));
                        continue;
                    }
                }
            }
            vx[r] = make_float4(dmg_div_rc(nx.x, k.beta, k.rb), dmg_div_rc(nx.y, k.beta, k.rb), dmg_div_rc(nx.z, k.beta, k.rb), dmg_div_rc(nx.w, k.beta, k.rb));
            vy[r] = make_float4(dmg_div_rc(ny.x, k.beta, k.rb), dmg_div_rc(ny.y, k.beta, k.rb), dmg_div_rc(ny.z, k.beta, k.rb), dmg_div_rc(ny.w, k.beta, k.rb));
        }
    }

    // store the interior: extended rows [T, EH-T), lanes 2..61, clipped to the launch's row range and to W
    const int xcol = X0 - TB_T + 4 * lane;          // unwrapped column
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
#pragma unroll
    for (int r = 0; r < R; r++) {
        int e = wave * R + r;
        int l = e0 + r;                             // the row itself (mirrored / clamped copies fail the range test)
        bool row_ok = (e >= TB_T) && (e < EH - TB_T) && (l >= a.out_lo) && (l < a.out_hi);
        if (row_ok && lane_ok) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f q0 = { vx[r].x, vy[r].x, vx[r].y, vy[r].y }, q1 = { vx[r].z, vy[r].z, vx[r].w, vy[r].w };
            v4f *dst = reinterpret_cast<v4f *>(a.vout + (size_t)l * W + xcol);
            __builtin_nontemporal_store(q0, dst);   // never re-read in this launch
            __builtin_nontemporal_store(q1, dst + 1);
        }
    }
}

template <int R, int NW>
static void dtb_launch(dtb_args a, int rows, hipStream_t stream)
{
    constexpr int OH = R * NW - 2 * TB_T;
    a.tiles_x = (a.W + TB_OW - 1) / TB_OW;
    a.tiles_y = (rows + OH - 1) / OH;
    hipLaunchKernelGGL((k_diffusion_tb<R, NW>), dim3((unsigned)(a.tiles_x * a.tiles_y)), dim3(NW * 64), 0, stream, a);
}

// host launcher: `n` (1..8) diffusion sweeps over local rows [lo, hi).  The caller flips o->vcur.
int ocean_launch_diffusion_tb(dmg_ocean *o, int lo, int hi, int n)
{
    if (hi <= lo) return DMG_OK;
    if (n < 1 || n > TB_T) return dmg_fail(DMG_EINVAL, "blocked diffusion: %d sweeps per launch (1..%d)", n, TB_T);
    dtb_args a;
    a.vin = reinterpret_cast<const float2 *>(o->v[o->vcur]);
    a.vout = reinterpret_cast<float2 *>(o->v[o->vcur ^ 1]);
    a.code = o->code;
    a.px = o->rt(RT_DIFPX); a.py = o->rt(RT_DIFPY); a.beta = o->rt(RT_DIFBETA); a.rb = o->rt(RT_DIFRB);
    a.W = o->grid.W; a.L = o->L;
    a.out_lo = lo; a.out_hi = hi;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    a.nsweeps = n;
    a.rowclass = o->rowclass;
    a.scalar_rhs = (o->prm.quirks & DMG_QUIRK_DIFFUSION_SCALAR_RHS) ? 1 : 0;
    // tile shape: every shape gives identical results.  56-row tiles (R = 7: 128 VGPRs with 16 B of spill, two workgroups
    // per CU) for launches of thousands of tiles, 48-row ones (R = 6, 122 VGPRs) for hundreds, 32-row ones below.
    // Measured, 50 sweeps at 16384x8192: R = 7 7.6 ms, R = 6 7.8 ms,
    // R = 4 10.3 ms, R = 8 13.5 ms (it needs ~150 VGPRs and spills at the 128 that two resident workgroups allow;
    // 6-wave workgroups of 8 or 10 rows per thread at 3 waves per SIMD: 11.5 / 10.9 ms);
    // one sweep per launch: 28.7 ms.  DMG_DTB_ROWS=4|6|7|8 forces a height for experiments.
    static const int forced = [] { const char *e = dmg_hook("DMG_DTB_ROWS"); int r = e ? atoi(e) : 0; return (r == 4 || r == 6 || r == 7 || r == 8) ? r : 0; }();
    const int tiles88 = ((a.W + TB_OW - 1) / TB_OW) * ((hi - lo + 47) / 48);
    const int R = forced ? forced : (tiles88 > 2000 ? 7 : (tiles88 > 400 ? 6 : 4));
    hipStream_t st = o->ctx->stream;
    if (R == 4) dtb_launch<4, 8>(a, hi - lo, st);
    else if (R == 7) dtb_launch<7, 8>(a, hi - lo, st);
    else if (R == 6) dtb_launch<6, 8>(a, hi - lo, st);
    else dtb_launch<8, 8>(a, hi - lo, st);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}
