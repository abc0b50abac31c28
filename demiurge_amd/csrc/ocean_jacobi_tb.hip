// Temporally blocked Jacobi: up to TB_T = 8 sweeps of the pressure stencil (reference
// src/filter/OceanCurrents.cpp:450-481) per launch, bit-identical to 8 single-sweep launches.
//
// Why: one sweep per launch is HBM-bound at 13 B/cell (ocean.hip).  Jacobi updates are order-independent per
// cell, so a workgroup can load a tile plus an 8-cell halo once, advance it 8 sweeps on chip (recomputing the
// halo redundantly) and write the interior once: ~15 B per cell per 8 sweeps.
//
// Tile geometry (one workgroup = 8 waves = 512 threads):
//     extended tile 256 columns x 128 rows;  wave w owns rows [16w, 16w+16), lane j owns columns [4j, 4j+4)
//     -> each thread keeps a 4 x 16 block of pressure AND of the right-hand side in registers (128 VGPRs)
//     output = the interior 240 x 112 cells (8 cells from every edge; stale values creep inwards one cell per sweep)
// Neighbour traffic per sweep:
//     east/west : adjacent lanes of the same wave -> DPP wave_shr:1 / wave_shl:1 (no LDS)
//     north/south between the 16-row blocks of adjacent waves -> 2 rows per wave through LDS (double-buffered by
//     sweep parity, one s_barrier per sweep)
// Canvas topology is applied when the extended tile is loaded: columns wrap (full longitude), rows beyond a pole
// are the same rows read mirrored and shifted by W/2 (offset(), src/Shader.h:86-95); rows beyond a non-pole edge
// are plain mirrors (CLAMP_TO_EDGE on a symmetric stencil).  Mirrored rows swap the N/S bits of their code.
#include "ocean_internal.h"

#define TB_T 8            // halo depth = max sweeps per launch
#define TB_R 16           // rows per thread
#define TB_WAVES 8
#define TB_EW 256         // extended tile width  (64 lanes x 4)
#define TB_EH (TB_R * TB_WAVES)
#define TB_OW (TB_EW - 2 * TB_T)      // 240
#define TB_OH (TB_EH - 2 * TB_T)      // 112

struct tb_args {
    const float *pin;
    float *pout;
    const float *b;
    const unsigned char *code;
    const float *cx, *cy, *beta;    // per local row
    int W, L;
    int out_lo, out_hi;             // local rows written by this launch
    int y_first, H;                 // grid row of local row 0; grid height
    int spole, npole;
    int nsweeps;                    // 1..TB_T
};

__device__ __forceinline__ float dpp_from_prev_lane(float v)   // lane i <- lane i-1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_next_lane(float v)   // lane i <- lane i+1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// same arithmetic as jacobi_cell() in ocean.hip (GLSL statement order, IEEE division)
__device__ __forceinline__ float tb_cell_fast(float pN, float pS, float pE, float pW, float b, float cx, float cy, float beta)
{
    return __fdiv_rn(__fsub_rn(__fadd_rn(__fdiv_rn(__fadd_rn(pW, pE), cx), __fdiv_rn(__fadd_rn(pS, pN), cy)), b), beta);
}
__device__ __forceinline__ float tb_cell(float pC, float pN, float pS, float pE, float pW, float b, unsigned c,
                                         float cx, float cy, float beta)
{
    if (c & 1u) pN = pC;
    if (c & 2u) pS = pC;
    if (c & 4u) pE = pC;
    if (c & 8u) pW = pC;
    float r = tb_cell_fast(pN, pS, pE, pW, b, cx, cy, beta);
    return (c & 16u) ? 0.0f : r;
}

__global__ __launch_bounds__(512) void k_jacobi_tb8(tb_args a)
{
    __shared__ float4 halo[2][TB_WAVES][2][64];     // [sweep parity][wave][0 = bottom row r=0, 1 = top row r=R-1][lane]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = a.W;
    const int X0 = blockIdx.x * TB_OW;                       // first output column of the tile
    const int Y0 = a.out_lo + blockIdx.y * TB_OH;            // first output local row of the tile

    // this lane's 4 columns (wrapped), plain and pole-shifted
    int col = X0 - TB_T + 4 * lane;
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;

    float4 p[TB_R], b[TB_R];
    unsigned code[TB_R];
    float rcx[TB_R], rcy[TB_R], rbeta[TB_R];
    int lrow[TB_R];                                           // local source row of each register row (for the store)
    unsigned slow_rows = 0;                                   // bit r: some lane of this wave has land/coast in row r

#pragma unroll
    for (int r = 0; r < TB_R; r++) {
        int e = wave * TB_R + r;                              // extended-tile row
        int y = a.y_first + Y0 - TB_T + e;                    // grid row (may fall outside the grid)
        bool flip = false, shift = false;
        if (y < 0) { y = -1 - y; flip = true; shift = a.spole != 0; }
        else if (y >= a.H) { y = 2 * a.H - 1 - y; flip = true; shift = a.npole != 0; }
        int l = y - a.y_first;
        l = min(max(l, 0), a.L - 1);                          // rows past the stored band only feed cells that are not written
        lrow[r] = flip ? -1 : l;                              // mirrored rows are never stored
        size_t off = (size_t)l * W + (shift ? colp : col);
        p[r] = *reinterpret_cast<const float4 *>(a.pin + off);
        b[r] = *reinterpret_cast<const float4 *>(a.b + off);
        unsigned c = *reinterpret_cast<const unsigned *>(a.code + off);
        if (flip) c = (c & 0xFCFCFCFCu) | ((c & 0x01010101u) << 1) | ((c & 0x02020202u) >> 1);   // swap N/S bits
        code[r] = c;
        rcx[r] = a.cx[l]; rcy[r] = a.cy[l]; rbeta[r] = a.beta[l];
        if (__builtin_amdgcn_ballot_w64((c & 0x1F1F1F1Fu) != 0)) slow_rows |= (1u << r);
    }
    slow_rows = __builtin_amdgcn_readfirstlane(slow_rows);

    for (int s = 0; s < a.nsweeps; s++) {
        // publish this wave's bottom and top rows (values before the sweep), fetch the neighbours' rows
        const int par = s & 1;
        halo[par][wave][0][lane] = p[0];
        halo[par][wave][1][lane] = p[TB_R - 1];
        __syncthreads();
        float4 south = (wave > 0) ? halo[par][wave - 1][1][lane] : p[0];                 // row below r = 0
        float4 north = (wave < TB_WAVES - 1) ? halo[par][wave + 1][0][lane] : p[TB_R - 1];   // row above r = R-1

        float4 prev = south;                                  // old values of row r-1
#pragma unroll
        for (int r = 0; r < TB_R; r++) {
            float4 C = p[r];
            float4 N = (r < TB_R - 1) ? p[r + 1] : north;
            float4 S = prev;
            float wl = dpp_from_prev_lane(C.w);               // west neighbour of C.x
            float er = dpp_from_next_lane(C.x);               // east neighbour of C.w
            float cx = rcx[r], cy = rcy[r], be = rbeta[r];
            float4 o;
            if (!(slow_rows & (1u << r))) {                   // wave-uniform: open ocean, no selects needed
                o.x = tb_cell_fast(N.x, S.x, C.y, wl, b[r].x, cx, cy, be);
                o.y = tb_cell_fast(N.y, S.y, C.z, C.x, b[r].y, cx, cy, be);
                o.z = tb_cell_fast(N.z, S.z, C.w, C.y, b[r].z, cx, cy, be);
                o.w = tb_cell_fast(N.w, S.w, er, C.z, b[r].w, cx, cy, be);
            } else {
                unsigned c = code[r];
                o.x = tb_cell(C.x, N.x, S.x, C.y, wl, b[r].x, c & 0xFFu, cx, cy, be);
                o.y = tb_cell(C.y, N.y, S.y, C.z, C.x, b[r].y, (c >> 8) & 0xFFu, cx, cy, be);
                o.z = tb_cell(C.z, N.z, S.z, C.w, C.y, b[r].z, (c >> 16) & 0xFFu, cx, cy, be);
                o.w = tb_cell(C.w, N.w, S.w, er, C.z, b[r].w, (c >> 24) & 0xFFu, cx, cy, be);
            }
            prev = C;
            p[r] = o;
        }
    }

    // store the interior: extended rows [T, EH-T), lanes 2..61, clipped to the launch's row range and to W
    const int xcol = X0 - TB_T + 4 * lane;                    // unwrapped column
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
#pragma unroll
    for (int r = 0; r < TB_R; r++) {
        int e = wave * TB_R + r;
        int l = lrow[r];
        bool row_ok = (e >= TB_T) && (e < TB_EH - TB_T) && (l >= a.out_lo) && (l < a.out_hi) &&
                      (l == Y0 - TB_T + e);                   // not a clamped duplicate
        if (row_ok && lane_ok) *reinterpret_cast<float4 *>(a.pout + (size_t)l * W + xcol) = p[r];
    }
}

// host launcher: `n` (1..8) sweeps over local rows [lo, hi); returns DMG_OK or an error.  The caller flips o->cur.
int ocean_launch_tb(dmg_ocean *o, int lo, int hi, int n)
{
    if (hi <= lo) return DMG_OK;
    tb_args a;
    a.pin = o->p[o->cur];
    a.pout = o->p[o->cur ^ 1];
    a.b = o->divw;
    a.code = o->code;
    a.cx = o->jac_cx; a.cy = o->jac_cy; a.beta = o->jac_beta;
    a.W = o->grid.W; a.L = o->L;
    a.out_lo = lo; a.out_hi = hi;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    a.nsweeps = n;
    dim3 grid((o->grid.W + TB_OW - 1) / TB_OW, (hi - lo + TB_OH - 1) / TB_OH);
    hipLaunchKernelGGL(k_jacobi_tb8, grid, dim3(512), 0, o->ctx->stream, a);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

// the temporally blocked kernel needs: full longitude (columns wrap), float4-aligned pole shift, a tile-wide row
bool ocean_tb_supported(const dmg_ocean *o)
{
    return o->topo.lonwrap && (o->grid.W % 8 == 0) && o->grid.W >= TB_EW;
}
