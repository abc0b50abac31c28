// Whole-grid stencil filters behind the generic entry points of include/demiurge_hip.h:
//   dmg_blur               Blur::Blur + Blur::step          reference src/filter/BlurMenu.cpp:24-117
//   dmg_temperature_steps  Temperature::run inner pass      reference src/filter/Temperature.cpp:56-150
//   dmg_erosion_pass       erosion/uplift pass of cpufilter reference src/filter/cpufilter.cpp:106-214
// One thread per cell, row-major coalesced.  The fractional-offset taps (blur13, the temperature laplacian's
// 1/cos(lat) x-offsets) go through a literal fp32 restatement of the GLSL offset() (reference src/Shader.h:82-98)
// and of the texture unit (GL_LINEAR / GL_NEAREST with CLAMP_TO_EDGE, reference src/Texture.cpp:29-33), so the
// texel each tap hits is the one the reference hits.  Row-constant transcendental factors (cos(lat)) come from the
// host in fp32, exactly as in the ocean kernels; the per-row insolation of the climate step is evaluated on the
// device (sin/cos/asin/acos/tan of device libm: a few ulp from glibc).
#include <memory>
#include <new>
#include "canvas.h"
#include <algorithm>

// ---------------------------------------------------------------------------------------------------------------
// blur13 (BlurMenu.cpp:41-62): 7 bilinear taps along `direction`, x offsets divided by cos(|lat|)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur13(canvas c, const float *__restrict__ src, float *__restrict__ dst,
                                                const float *__restrict__ phifactor, float dirx, float diry)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = c.row0 + (int)blockIdx.y;
    if (x >= c.W) return;
    const float k1 = 1.411764705882353f, k2 = 3.2941176470588234f, k3 = 5.176470588235294f;
    const float w0 = 0.1964825501511404f, w1 = 0.2969069646728344f, w2 = 0.09447039785044732f, w3 = 0.010381362401148057f;
    float pf = phifactor[y];
    float o1x = __fdiv_rn(__fmul_rn(k1, dirx), pf), o2x = __fdiv_rn(__fmul_rn(k2, dirx), pf), o3x = __fdiv_rn(__fmul_rn(k3, dirx), pf);
    float o1y = __fmul_rn(k1, diry), o2y = __fmul_rn(k2, diry), o3y = __fmul_rn(k3, diry);
    float sx = d_stx(c, x), sy = d_sty(c, y);
    float color = 0.0f, px, py;
    color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, sx, sy), w0));
    d_offset(c, sx, sy, o1x, o1y, &px, &py);   color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, px, py), w1));
    d_offset(c, sx, sy, -o1x, -o1y, &px, &py); color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, px, py), w1));
    d_offset(c, sx, sy, o2x, o2y, &px, &py);   color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, px, py), w2));
    d_offset(c, sx, sy, -o2x, -o2y, &px, &py); color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, px, py), w2));
    d_offset(c, sx, sy, o3x, o3y, &px, &py);   color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, px, py), w3));
    d_offset(c, sx, sy, -o3x, -o3y, &px, &py); color = __fadd_rn(color, __fmul_rn(d_fetch_linear(src, c.W, c.H, px, py), w3));
    dst[(size_t)y * c.W + x] = color;
}

// The same pass when the direction lies along one axis -- the only two directions Blur::step uses, (0, r) and (r, 0)
// (BlurMenu.cpp:97-117).  k_blur13 evaluates offset() and the texture unit seven times per cell, ~1000 instructions, and is
// bound by them (83 us for 4096x2048, 0.8 TB/s); along an axis most of that is the same for every tap, and some of it for
// every cell of the row:
//   DIR 0, (r, 0): dp.y = +-(k * 0) / H = +-0, so every tap keeps py = st.y: no pole branch, and the two rows, their weight
//          and their addresses are the cell's own.  The x offsets (k r / cos|lat|) / W are three divisions per ROW: threads
//          0..2 of the block (one block = one row segment) leave them in LDS.
//   DIR 1, (0, r): dp.x = +-0, so px is the cell's own unless the tap crosses a pole, where it is the cell's px mirrored
//          (Shader.h:88,93): two candidate columns pairs and weights per cell, chosen per tap.  The y offsets (k r) / H do
//          not even depend on the row.
// Every operation that remains is the one k_blur13 does, in its order, on the same values (negating a dividend negates the
// quotient; adding a signed zero to a positive coordinate returns it): bit-identical, and tests/test_gpu_filters.py holds the
// two against each other as well as against the goldens.
struct blur_x { int i0, i1; float wx; };
__device__ __forceinline__ blur_x blur_xpart(float px, int W)
{
    const float fu = __fsub_rn(__fmul_rn(px, (float)W), 0.5f), flu = floorf(fu);
    blur_x r;
    r.wx = __fsub_rn(fu, flu);
    r.i0 = min(max((int)flu, 0), W - 1); r.i1 = min(max((int)flu + 1, 0), W - 1);
    return r;
}
struct blur_y { int j0, j1; float wy; };
__device__ __forceinline__ blur_y blur_ypart(float py, int H)
{
    const float fv = __fsub_rn(__fmul_rn(py, (float)H), 0.5f), flv = floorf(fv);
    blur_y r;
    r.wy = __fsub_rn(fv, flv);
    r.j0 = min(max((int)flv, 0), H - 1); r.j1 = min(max((int)flv + 1, 0), H - 1);
    return r;
}
__device__ __forceinline__ float blur_tap(const float *__restrict__ t, int W, const blur_x &bx, const blur_y &by)
{
    const float *r0 = t + (size_t)by.j0 * W, *r1 = t + (size_t)by.j1 * W;
    const float t00 = r0[bx.i0], t10 = r0[bx.i1], t01 = r1[bx.i0], t11 = r1[bx.i1];
    const float a = fmaf(bx.wx, __fsub_rn(t10, t00), t00);       // (d_fetch_linear)
    const float b = fmaf(bx.wx, __fsub_rn(t11, t01), t01);
    return fmaf(by.wy, __fsub_rn(b, a), a);
}
// (an offset longer than the globe is tall crosses both poles: the literal form, kept out of line)
__device__ __noinline__ float blur_tap_literal(canvas c, const float *__restrict__ src, float sx, float sy, float dpx, float dpy)
{
    float qx, qy;
    d_offset(c, sx, sy, dpx, dpy, &qx, &qy);
    return d_fetch_linear(src, c.W, c.H, qx, qy);
}
// BLUR_CPT cells per thread, 256 columns apart.  Measured at 4096x2048: one cell per thread 58 us (r, 0) / 70 us (0, r), four
// cells per thread 61 / 74 us -- the pass is not bound by its instructions any more (358 of k_blur13's 1093 are left in the
// (r, 0) form) nor by a block's latency chain, but by its 28 four-byte loads per cell through the vector memory path
// (940 MB of L1 traffic per pass; the HBM side is 67 MB)
#define BLUR_CPT 1
template <int DIR>
__global__ __launch_bounds__(256) void k_blur13_axis(canvas c, const float *__restrict__ src, float *__restrict__ dst,
                                                     const float *__restrict__ phifactor, float r)
{
    __shared__ float s_off[3];
    const int y = c.row0 + (int)blockIdx.y;
    if (threadIdx.x < 3) {
        const float k = threadIdx.x == 0 ? 1.411764705882353f : (threadIdx.x == 1 ? 3.2941176470588234f : 5.176470588235294f);
        float o = __fmul_rn(k, r);
        if (DIR == 0) o = __fdiv_rn(o, phifactor[y]);
        s_off[threadIdx.x] = __fdiv_rn(o, DIR == 0 ? (float)c.W : (float)c.H);
    }
    __syncthreads();
    const float w0 = 0.1964825501511404f, w[3] = { 0.2969069646728344f, 0.09447039785044732f, 0.010381362401148057f };
    const float sy = d_sty(c, y);
    const float off[3] = { s_off[0], s_off[1], s_off[2] };
    float color[BLUR_CPT];
    if (DIR == 0) {
        const blur_y by = blur_ypart(sy, c.H);
#pragma unroll
        for (int j = 0; j < BLUR_CPT; j++) {
            const int x = min((int)(blockIdx.x * (256 * BLUR_CPT) + 256 * j + threadIdx.x), c.W - 1);    // (a column past the edge repeats the last: not stored)
            const float sx = d_stx(c, x);
            color[j] = __fadd_rn(0.0f, __fmul_rn(d_fetch_linear(src, c.W, c.H, sx, sy), w0));
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int sgn = 0; sgn < 2; sgn++) {
                    float px = __fadd_rn(sx, sgn ? -off[k] : off[k]);
                    if (c.lonwrap) { px = d_mod(__fadd_rn(px, 1.0f), 1.0f); px = d_mod(__fadd_rn(px, 1.0f), 1.0f); }
                    color[j] = __fadd_rn(color[j], __fmul_rn(blur_tap(src, c.W, blur_xpart(px, c.W), by), w[k]));
                }
        }
    } else {
        // the rows, their weights and whether a tap crosses a pole are the same for the whole row
        blur_y by[6];
        int crossed[6];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int sgn = 0; sgn < 2; sgn++) {
                float py = __fadd_rn(sy, sgn ? -off[k] : off[k]);
                int cr = 0;
                if (c.spole && py < 0.0f) { py = -py; cr++; }
                if (c.npole && py > 1.0f) { py = __fsub_rn(2.0f, py); cr++; }
                crossed[2 * k + sgn] = cr;
                by[2 * k + sgn] = blur_ypart(py, c.H);
            }
        const bool any_crossed = (crossed[0] | crossed[1] | crossed[2] | crossed[3] | crossed[4] | crossed[5]) != 0;
#pragma unroll
        for (int j = 0; j < BLUR_CPT; j++) {
            const int x = min((int)(blockIdx.x * (256 * BLUR_CPT) + 256 * j + threadIdx.x), c.W - 1);
            const float sx = d_stx(c, x);
            color[j] = __fadd_rn(0.0f, __fmul_rn(d_fetch_linear(src, c.W, c.H, sx, sy), w0));
            float px1 = sx;
            if (c.lonwrap) px1 = d_mod(__fadd_rn(px1, 1.0f), 1.0f);
            const blur_x plain = blur_xpart(c.lonwrap ? d_mod(__fadd_rn(px1, 1.0f), 1.0f) : px1, c.W);
            blur_x mirrored = plain;
            if (any_crossed) {                                   // (the same for every cell of the row: no divergence)
                float px = d_pole_lon(__fadd_rn(__fmul_rn(px1, __fsub_rn(c.c3, c.c2)), c.c2));
                px = __fdiv_rn(__fsub_rn(px, c.c2), __fsub_rn(c.c3, c.c2));
                if (c.lonwrap) px = d_mod(__fadd_rn(px, 1.0f), 1.0f);
                mirrored = blur_xpart(px, c.W);
            }
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int sgn = 0; sgn < 2; sgn++) {
                    const int t = 2 * k + sgn;
                    float v;
                    if (crossed[t] == 2) {
                        const float kk = k == 0 ? 1.411764705882353f : (k == 1 ? 3.2941176470588234f : 5.176470588235294f);
                        v = blur_tap_literal(c, src, sx, sy, sgn ? -0.0f : 0.0f, sgn ? -__fmul_rn(kk, r) : __fmul_rn(kk, r));
                    } else v = blur_tap(src, c.W, crossed[t] ? mirrored : plain, by[t]);
                    color[j] = __fadd_rn(color[j], __fmul_rn(v, w[k]));
                }
        }
    }
#pragma unroll
    for (int j = 0; j < BLUR_CPT; j++) {
        const int x = (int)(blockIdx.x * (256 * BLUR_CPT) + 256 * j + threadIdx.x);
        if (x < c.W) dst[(size_t)y * c.W + x] = color[j];
    }
}

// Round 5: the same two passes with what is the same for a whole row kept in scalar registers.  profiles/r04_kernel_counters.txt
// had the axis forms bound by their vector instructions (236 / 409 per cell): most of them computed, in every lane, values that
// are the row's -- the tap rows, their weights and the pole crossings of the (0, r) pass, the two rows and the weight of the
// (r, 0) pass -- and 64-bit addresses for 28 loads per cell.  Here six threads compute the row's part once (the same operations
// on the same values), every lane reads it back through LDS and v_readfirstlane, and a tap's four loads are two scalar row bases
// plus the cell's two column offsets.  What stays per cell is what depends on x: the column pair and its weight (once for the
// (0, r) pass, per tap for the (r, 0) pass) and the three lerps per tap.  Bit-identical to k_blur13_axis and k_blur13
// (tests/test_gpu_filters.py holds the three against each other and against the goldens).
struct blur_row { int j0, j1, crossed; float wy; };
__device__ __forceinline__ blur_row blur_row_uniform(const blur_row &r)
{
    blur_row u;
    u.j0 = __builtin_amdgcn_readfirstlane(r.j0); u.j1 = __builtin_amdgcn_readfirstlane(r.j1);
    u.crossed = __builtin_amdgcn_readfirstlane(r.crossed);
    u.wy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r.wy)));
    return u;
}
__device__ __forceinline__ float blur_tap_rows(const float *__restrict__ r0, const float *__restrict__ r1, const blur_x &bx, float wy)
{
    const unsigned o0 = (unsigned)bx.i0 * 4u, o1 = (unsigned)bx.i1 * 4u;
    const char *b0 = reinterpret_cast<const char *>(r0), *b1 = reinterpret_cast<const char *>(r1);
    const float t00 = *reinterpret_cast<const float *>(b0 + o0), t10 = *reinterpret_cast<const float *>(b0 + o1);
    const float t01 = *reinterpret_cast<const float *>(b1 + o0), t11 = *reinterpret_cast<const float *>(b1 + o1);
    const float a = fmaf(bx.wx, __fsub_rn(t10, t00), t00);       // (d_fetch_linear)
    const float b = fmaf(bx.wx, __fsub_rn(t11, t01), t01);
    return fmaf(wy, __fsub_rn(b, a), a);
}
template <int DIR>
__global__ __launch_bounds__(256) void k_blur13_axis2(canvas c, const float *__restrict__ src, float *__restrict__ dst,
                                                      const float *__restrict__ phifactor, float r)
{
    __shared__ float s_off[3];
    __shared__ blur_row s_row[7];                       // [0..5]: the taps' rows (DIR 1); [6]: the cell's own rows
    const int y = c.row0 + (int)blockIdx.y, W = c.W;
    const float sy = d_sty(c, y);
    if (threadIdx.x < 3) {
        const float k = threadIdx.x == 0 ? 1.411764705882353f : (threadIdx.x == 1 ? 3.2941176470588234f : 5.176470588235294f);
        float o = __fmul_rn(k, r);
        if (DIR == 0) o = __fdiv_rn(o, phifactor[y]);
        o = __fdiv_rn(o, DIR == 0 ? (float)c.W : (float)c.H);
        s_off[threadIdx.x] = o;
        if (DIR == 1) {
#pragma unroll
            for (int sgn = 0; sgn < 2; sgn++) {
                float py = __fadd_rn(sy, sgn ? -o : o);
                int cr = 0;
                if (c.spole && py < 0.0f) { py = -py; cr++; }
                if (c.npole && py > 1.0f) { py = __fsub_rn(2.0f, py); cr++; }
                const blur_y by = blur_ypart(py, c.H);
                s_row[2 * threadIdx.x + sgn] = blur_row{ by.j0, by.j1, cr, by.wy };
            }
        }
    } else if (threadIdx.x == 3) {
        const blur_y by = blur_ypart(sy, c.H);
        s_row[6] = blur_row{ by.j0, by.j1, 0, by.wy };
    }
    __syncthreads();
    const float w0 = 0.1964825501511404f, w[3] = { 0.2969069646728344f, 0.09447039785044732f, 0.010381362401148057f };
    const blur_row own = blur_row_uniform(s_row[6]);
    const float *own0 = src + (size_t)own.j0 * W, *own1 = src + (size_t)own.j1 * W;
    const int x = min((int)(blockIdx.x * 256 + threadIdx.x), W - 1);        // (a column past the edge repeats the last: not stored)
    const float sx = d_stx(c, x);
    // the cell's own tap: d_fetch_linear(src, sx, sy)
    float color = __fadd_rn(0.0f, __fmul_rn(blur_tap_rows(own0, own1, blur_xpart(sx, W), own.wy), w0));
    if (DIR == 0) {
        float off[3];
#pragma unroll
        for (int k = 0; k < 3; k++) off[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_off[k])));
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int sgn = 0; sgn < 2; sgn++) {
                float px = __fadd_rn(sx, sgn ? -off[k] : off[k]);
                if (c.lonwrap) { px = d_mod(__fadd_rn(px, 1.0f), 1.0f); px = d_mod(__fadd_rn(px, 1.0f), 1.0f); }
                color = __fadd_rn(color, __fmul_rn(blur_tap_rows(own0, own1, blur_xpart(px, W), own.wy), w[k]));
            }
    } else {
        blur_row by[6];
        int any_crossed = 0;
#pragma unroll
        for (int t = 0; t < 6; t++) { by[t] = blur_row_uniform(s_row[t]); any_crossed |= by[t].crossed; }
        float px1 = sx;
        if (c.lonwrap) px1 = d_mod(__fadd_rn(px1, 1.0f), 1.0f);
        const blur_x plain = blur_xpart(c.lonwrap ? d_mod(__fadd_rn(px1, 1.0f), 1.0f) : px1, W);
        blur_x mirrored = plain;
        if (any_crossed) {                                   // (the same for every cell of the row: no divergence)
            float px = d_pole_lon(__fadd_rn(__fmul_rn(px1, __fsub_rn(c.c3, c.c2)), c.c2));
            px = __fdiv_rn(__fsub_rn(px, c.c2), __fsub_rn(c.c3, c.c2));
            if (c.lonwrap) px = d_mod(__fadd_rn(px, 1.0f), 1.0f);
            mirrored = blur_xpart(px, W);
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int sgn = 0; sgn < 2; sgn++) {
                const int t = 2 * k + sgn;
                float v;
                if (by[t].crossed == 2) {
                    const float kk = k == 0 ? 1.411764705882353f : (k == 1 ? 3.2941176470588234f : 5.176470588235294f);
                    v = blur_tap_literal(c, src, sx, sy, sgn ? -0.0f : 0.0f, sgn ? -__fmul_rn(kk, r) : __fmul_rn(kk, r));
                } else
                    v = blur_tap_rows(src + (size_t)by[t].j0 * W, src + (size_t)by[t].j1 * W, by[t].crossed ? mirrored : plain, by[t].wy);
                color = __fadd_rn(color, __fmul_rn(v, w[k]));
            }
    }
    if ((int)(blockIdx.x * 256 + threadIdx.x) < W) dst[(size_t)y * W + x] = color;
}

// Round 6: the (0, r) pass streamed down the columns.  Along y every tap of a cell reads the SAME column pair with the same weight
// (blur_x `plain`; the cell's own tap has its own pair, from st.x before the wraps), and a tap's value is
//     fma(wy, b - a, a)   with   a = fma(wx, T[j0][i1] - T[j0][i0], T[j0][i0]),   b = the same on row j1:
// a and b are values of a (source row, column) alone -- the row's horizontal lerp.  k_blur13_axis2<1> works each of them out once per
// tap that reads it: 28 loads and 14 lerps per cell, which is what bound it (200 vector instructions and 28 trips through the vector
// memory path per cell).  Here a thread owns a column and walks down `rows` rows of it; every source row's lerp is worked out ONCE --
// four loads, two lerps -- and kept in a ring of the thread's own in LDS (dynamic row index; a thread reads only what it wrote: no
// barrier on the data), and a tap is two LDS reads and the vertical lerp.  Same operations on the same operands in the same order
// as k_blur13 (tests/test_gpu_filters.py holds it against k_blur13_axis2, k_blur13 and the goldens).  What is the same for a whole
// row -- the taps' rows, weights, pole crossings -- is worked out by 56 threads per chunk of eight rows with the statements of
// k_blur13_axis2 and handed round through LDS.  A tap that crosses a pole (its rows are mirrored and its columns the mirrored pair)
// is taken the old way, straight from memory; RING rows must hold a row's taps: 2 K + 8 of them, K = ceil(5.18 r) + 2.
#define BLUR_YT_CHUNK 8
#define BLUR_YT_RO 16          // ring of the cell's own tap: rows y - 1 .. y + 1 and a chunk
// a source row's horizontal lerps, CHUNK rows at a time: rows [*jfill, jneed] into ring (loads of a row past jneed repeat jneed's)
template <int NR, int BT>
__device__ __forceinline__ void blur_yt_fill(float (*ring)[BT], const float *__restrict__ src, int W, unsigned o0, unsigned o1, float wx,
                                             int *jfill, int jneed, int tid)
{
    while (*jfill <= jneed) {
        float t0[BLUR_YT_CHUNK], t1[BLUR_YT_CHUNK];
#pragma unroll
        for (int q = 0; q < BLUR_YT_CHUNK; q++) {
            const int j = min(*jfill + q, jneed);
            const char *row = reinterpret_cast<const char *>(src + (size_t)j * W);
            t0[q] = *reinterpret_cast<const float *>(row + o0); t1[q] = *reinterpret_cast<const float *>(row + o1);
        }
#pragma unroll
        for (int q = 0; q < BLUR_YT_CHUNK; q++) {       // (a row past jneed goes to the spare slot NR: no branch, so every load is waited for here)
            const int j = *jfill + q;
            ring[j <= jneed ? (j & (NR - 1)) : NR][tid] = fmaf(wx, __fsub_rn(t1[q], t0[q]), t0[q]);      // (blur_tap_rows' a and b)
        }
        *jfill = min(*jfill + BLUR_YT_CHUNK, jneed + 1);
    }
}
// a tap's rows as k_blur13_axis2 has them, and where the ring keeps those rows (bytes from the ring's start to the row, for a row
// that crosses no pole); any: some tap of this output row crosses a pole (kept with the cell's own rows, t = 6)
struct alignas(32) blur_yrec { int j0, j1, crossed, pad; float wy; int o0, o1, any; };
// ... and the part of a row's seven records the common case reads, in one 64-byte line (ONE s_load_dwordx16 per row): per tap its
// weight and its two ring offsets in sixteen bits each (a ring is at most 32 KB), then `any`
struct alignas(64) blur_yfast { struct { float wy; unsigned o01; } t[7]; int any, pad; };
static_assert(sizeof(blur_yfast) == 64 && sizeof(blur_yrec) == 32, "one cache line; two 16-byte halves");
// What is the same for a whole row of the (0, r) pass -- the taps' rows, weights, pole crossings: what k_blur13_axis2<1>'s first four
// threads work out per row, statement for statement -- once per pass into a table the main kernel reads with scalar loads:
// tab[(y - c.row0) * 7 + t], t = 0..5 the taps in k_blur13's order, 6 the cell's own rows.  ring / pitch: rows and bytes per row of
// the main kernel's LDS ring.
__global__ void k_blur13_yrec(canvas c, float r, int nrows, int ring, int pitch, blur_yrec *__restrict__ tab, blur_yfast *__restrict__ fast)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * 7) return;
    const int yy = c.row0 + i / 7, t = i % 7, H = c.H;
    const float sy = d_sty(c, yy);
    blur_yrec rr = {};
    int any = 0;
#pragma unroll
    for (int u = 0; u < 6; u++) {                 // (t = 6 looks at all six taps for `any`; the others at their own)
        if (t != 6 && t != u) continue;
        const int k = u >> 1, sgn = u & 1;
        const float kk = k == 0 ? 1.411764705882353f : (k == 1 ? 3.2941176470588234f : 5.176470588235294f);
        float o = __fmul_rn(kk, r);
        o = __fdiv_rn(o, (float)H);
        float py = __fadd_rn(sy, sgn ? -o : o);
        int cr = 0;
        if (c.spole && py < 0.0f) { py = -py; cr++; }
        if (c.npole && py > 1.0f) { py = __fsub_rn(2.0f, py); cr++; }
        any |= cr;
        if (t == u) {
            const blur_y by = blur_ypart(py, H);
            rr.j0 = by.j0; rr.j1 = by.j1; rr.crossed = cr; rr.wy = by.wy;
            rr.o0 = (by.j0 & (ring - 1)) * pitch; rr.o1 = (by.j1 & (ring - 1)) * pitch;
        }
    }
    if (t == 6) {
        const blur_y by = blur_ypart(sy, H);
        rr.j0 = by.j0; rr.j1 = by.j1; rr.crossed = 0; rr.wy = by.wy;
        rr.o0 = (by.j0 & (BLUR_YT_RO - 1)) * pitch; rr.o1 = (by.j1 & (BLUR_YT_RO - 1)) * pitch;
        rr.any = any;
    }
    tab[i] = rr;
    fast[i / 7].t[t].wy = rr.wy;
    fast[i / 7].t[t].o01 = (unsigned)rr.o0 | ((unsigned)rr.o1 << 16);
    if (t == 6) fast[i / 7].any = any;
}
template <int RING, int BT>
__global__ __launch_bounds__(BT) void k_blur13_ytile(canvas c, const float *__restrict__ src, float *__restrict__ dst, float r, int nrows, int rows, int K,
                                                     const blur_yrec *__restrict__ tab, const blur_yfast *__restrict__ fast)
{
    static_assert(RING * BT * 4 <= 65536, "ring offsets in sixteen bits");
    __shared__ float s_p[RING + 1][BT];                  // lerps with the taps' column pair, row j at j & (RING - 1) (+ a spare slot)
    __shared__ float s_o[BLUR_YT_RO + 1][BT];            // ... with the cell's own column pair
    const int W = c.W, H = c.H, tid = (int)threadIdx.x;
    const int y0 = c.row0 + (int)blockIdx.y * rows, yend = min(y0 + rows, c.row0 + nrows);
    const int x = min((int)(blockIdx.x * BT) + tid, W - 1);          // (a column past the edge repeats the last: not stored)
    const bool store = (int)(blockIdx.x * BT) + tid < W;
    const float sx = d_stx(c, x);
    const blur_x xo = blur_xpart(sx, W);
    float px1 = sx;
    if (c.lonwrap) px1 = d_mod(__fadd_rn(px1, 1.0f), 1.0f);
    const blur_x plain = blur_xpart(c.lonwrap ? d_mod(__fadd_rn(px1, 1.0f), 1.0f) : px1, W);
    const float w0 = 0.1964825501511404f, w[3] = { 0.2969069646728344f, 0.09447039785044732f, 0.010381362401148057f };
    int jfill = max(y0 - K, 0), jfill_o = max(y0 - 1, 0);
    const char *ringp = reinterpret_cast<const char *>(&s_p[0][tid]), *ringo = reinterpret_cast<const char *>(&s_o[0][tid]);
    for (int yc = y0; yc < yend; yc += BLUR_YT_CHUNK) {
        const int ylast = min(yc + BLUR_YT_CHUNK, yend) - 1;
        // the source rows the chunk needs and the rings do not hold yet: two loads and a lerp each, a chunk's worth in flight
        blur_yt_fill<RING, BT>(s_p, src, W, (unsigned)plain.i0 * 4u, (unsigned)plain.i1 * 4u, plain.wx, &jfill, min(H - 1, ylast + K), tid);
        blur_yt_fill<BLUR_YT_RO, BT>(s_o, src, W, (unsigned)xo.i0 * 4u, (unsigned)xo.i1 * 4u, xo.wx, &jfill_o, min(H - 1, ylast + 1), tid);
        // A row's step is a chain of latencies -- its records from the scalar cache, then its ring reads: the next row's line is asked
        // for before this row's ring reads, so the two waits are one
        blur_yfast cur = fast[yc - c.row0];
        for (int y = yc; y <= ylast; y++) {
            const blur_yfast nxt = fast[min(y + 1, ylast) - c.row0];
            const blur_yfast q = cur;
            cur = nxt;
            float color;
            if (q.any == 0) {
                // no tap of this row crosses a pole: fourteen ring reads at the records' offsets, then the taps in k_blur13's order
                float a[7], b[7];
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const char *ring = t == 6 ? ringo : ringp;
                    a[t] = *reinterpret_cast<const float *>(ring + (q.t[t].o01 & 0xffffu)); b[t] = *reinterpret_cast<const float *>(ring + (q.t[t].o01 >> 16));
                }
                color = __fadd_rn(0.0f, __fmul_rn(fmaf(q.t[6].wy, __fsub_rn(b[6], a[6]), a[6]), w0));
#pragma unroll
                for (int t = 0; t < 6; t++) color = __fadd_rn(color, __fmul_rn(fmaf(q.t[t].wy, __fsub_rn(b[t], a[t]), a[t]), w[t >> 1]));
            } else {
                const blur_yrec *rec = tab + (size_t)(y - c.row0) * 7;
                const float a0 = *reinterpret_cast<const float *>(ringo + rec[6].o0), b0 = *reinterpret_cast<const float *>(ringo + rec[6].o1);
                color = __fadd_rn(0.0f, __fmul_rn(fmaf(rec[6].wy, __fsub_rn(b0, a0), a0), w0));
                float px = d_pole_lon(__fadd_rn(__fmul_rn(px1, __fsub_rn(c.c3, c.c2)), c.c2));
                px = __fdiv_rn(__fsub_rn(px, c.c2), __fsub_rn(c.c3, c.c2));
                if (c.lonwrap) px = d_mod(__fadd_rn(px, 1.0f), 1.0f);
                const blur_x mirrored = blur_xpart(px, W);
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int sgn = 0; sgn < 2; sgn++) {
                        const int t = 2 * k + sgn;
                        const blur_yrec rt = rec[t];
                        float v;
                        if (rt.crossed == 2) {
                            const float kk = k == 0 ? 1.411764705882353f : (k == 1 ? 3.2941176470588234f : 5.176470588235294f);
                            v = blur_tap_literal(c, src, sx, d_sty(c, y), sgn ? -0.0f : 0.0f, sgn ? -__fmul_rn(kk, r) : __fmul_rn(kk, r));
                        } else if (rt.crossed)
                            v = blur_tap_rows(src + (size_t)rt.j0 * W, src + (size_t)rt.j1 * W, mirrored, rt.wy);
                        else {
                            const float a = *reinterpret_cast<const float *>(ringp + rt.o0), b = *reinterpret_cast<const float *>(ringp + rt.o1);
                            v = fmaf(rt.wy, __fsub_rn(b, a), a);
                        }
                        color = __fadd_rn(color, __fmul_rn(v, w[k]));
                    }
            }
            if (store) dst[(size_t)y * W + x] = color;
        }
    }
}

// one pass of blur13 in direction (dirx, diry): the axis forms where they apply
// yrec: room for a blur_yfast and 7 blur_yrec per row of the launch (BLUR_YREC_BYTES), or null: the (0, r) pass by k_blur13_axis2<1>
#define BLUR_YREC_BYTES(nrows) ((size_t)(nrows) * (sizeof(blur_yfast) + 7 * sizeof(blur_yrec)))
static void launch_blur13(hipStream_t st, dim3 g, const canvas &c, const float *src, float *dst, const float *pf, float dirx, float diry, void *yrec = nullptr)
{
    static const bool literal = dmg_hook("DMG_BLUR_LITERAL") != nullptr;       // tests: the general kernel for every pass
    const dim3 ga((unsigned)((c.W + 256 * BLUR_CPT - 1) / (256 * BLUR_CPT)), g.y);
    static const bool axis_v1 = dmg_hook("DMG_BLUR_AXIS_V1") != nullptr;       // tests, A/B: the round-3 axis kernels
    static const bool no_ytile = dmg_hook("DMG_BLUR_NO_YTILE") != nullptr;     // tests, A/B: the (0, r) pass by k_blur13_axis2<1> (round 5's form)
    if (!literal && !axis_v1 && !no_ytile && yrec && dirx == 0.0f && diry > 0.0f) {
        // the (0, r) pass streamed down the columns where a ring of 16 / 32 / 64 rows holds a row's taps (r up to 5.2)
        const int K = (int)ceilf(5.1765f * diry) + 1, need = 2 * K + BLUR_YT_CHUNK, nrows = (int)g.y;
        static const int rows_hook = [] { const char *e = dmg_hook("DMG_BLUR_YT_ROWS"); return e ? atoi(e) : 0; }();   // tests: every ring size on small grids
        // (below two million cells a pass is its launches: the records' pre-kernel would be a seventh more of them -- 1024 x 512, config 1)
        if (need <= 64 && nrows >= 1 && ((size_t)nrows * c.W >= ((size_t)1 << 21) || rows_hook)) {
            const int BT = need <= 32 ? 256 : 128;
            const int bx = (c.W + BT - 1) / BT;
            // rows per block: enough blocks to fill the chip, as tall as that allows (a block re-reads 2 K rows)
            int rows = (int)(((long long)nrows * bx + 4095) / 4096);
            rows = std::min(std::max((rows + BLUR_YT_CHUNK - 1) / BLUR_YT_CHUNK * BLUR_YT_CHUNK, 2 * BLUR_YT_CHUNK), 256);
            if (rows_hook >= BLUR_YT_CHUNK) rows = rows_hook / BLUR_YT_CHUNK * BLUR_YT_CHUNK;
            const dim3 gt((unsigned)bx, (unsigned)((nrows + rows - 1) / rows));
            // (a block reads rows + 2 K source rows: on a small grid with a large radius -- 4096 x 2048 from r = 1.2 -- that is
            // more than twice its rows, and k_blur13_axis2<1> is the faster one: tools/blur_pass_probe.py)
            if (2 * K <= rows || rows_hook) {
            blur_yfast *fast = (blur_yfast *)yrec;                      // (a pool block: 256-byte aligned)
            blur_yrec *tab = (blur_yrec *)(fast + nrows);
            hipLaunchKernelGGL(k_blur13_yrec, dim3((unsigned)((nrows * 7 + 255) / 256)), dim3(256), 0, st, c, diry, nrows, need <= 16 ? 16 : (need <= 32 ? 32 : 64), BT * 4, tab, fast);
            if (need <= 16) hipLaunchKernelGGL((k_blur13_ytile<16, 256>), gt, dim3(256), 0, st, c, src, dst, diry, nrows, rows, K, tab, fast);
            else if (need <= 32) hipLaunchKernelGGL((k_blur13_ytile<32, 256>), gt, dim3(256), 0, st, c, src, dst, diry, nrows, rows, K, tab, fast);
            else hipLaunchKernelGGL((k_blur13_ytile<64, 128>), gt, dim3(128), 0, st, c, src, dst, diry, nrows, rows, K, tab, fast);
            return;
            }
        }
    }
    if (!literal && !axis_v1 && diry == 0.0f && dirx > 0.0f) hipLaunchKernelGGL(k_blur13_axis2<0>, ga, dim3(256), 0, st, c, src, dst, pf, dirx);
    else if (!literal && !axis_v1 && dirx == 0.0f && diry > 0.0f) hipLaunchKernelGGL(k_blur13_axis2<1>, ga, dim3(256), 0, st, c, src, dst, pf, diry);
    else if (!literal && diry == 0.0f && dirx > 0.0f) hipLaunchKernelGGL(k_blur13_axis<0>, ga, dim3(256), 0, st, c, src, dst, pf, dirx);
    else if (!literal && dirx == 0.0f && diry > 0.0f) hipLaunchKernelGGL(k_blur13_axis<1>, ga, dim3(256), 0, st, c, src, dst, pf, diry);
    else hipLaunchKernelGGL(k_blur13, g, dim3(256), 0, st, c, src, dst, pf, dirx, diry);
}

// Blur::Blur radius schedule (BlurMenu.cpp:27,79-94), fp32 with the reference's implicit double sqrt
static std::vector<float> blur_rlist(float radius)
{
    std::vector<float> rl;
    radius = radius / 2;
    float R = radius * radius / 2;
    float i = 1.0f, incrementer = 0.5f;
    if (R < 3) {
        float k = (float)(1 / sqrt((double)(55 / R)));
        incrementer = k;
        i = k;
    }
    while (R >= i * i) {
        R -= i * i;
        rl.push_back(i);
        i += incrementer;
    }
    if (R > 0.0f) rl.push_back((float)sqrt((double)R));
    std::sort(rl.begin(), rl.end());
    return rl;
}

// ---------------------------------------------------------------------------------------------------------------
// climate step (Temperature.cpp:56-134)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// QDay(phi, M) (:69-95) -> absorbed solar radiation of the row, ASR = (1-0.30)*QDay (:102,:122)
// (one launch for a batch of steps: blockIdx.y = the step within the batch; M = (float)(M_PI*2/15000*i), the host's
// expression in the host's double arithmetic, :146)
#define TEMP_ROWS_BATCH 64
// the orbit and the surface: every literal of the shader's preamble (Temperature.cpp:56-95) and of :102, as the host hands them over
struct temp_orbit { float eccentricity, sin_gamma, omega2, S0, one_minus_albedo; int steps_per_year, clamp_acos; };
__global__ void k_temperature_rows(canvas c, const float *__restrict__ lat, int first_step, temp_orbit ob, float *__restrict__ asr_batch)
{
    int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= c.H) return;
    const float M = (float)(M_PI * 2 / ob.steps_per_year * (first_step + (int)blockIdx.y));
    float *asr = asr_batch + (size_t)blockIdx.y * c.H;
    float phi = lat[y];
    const float eccentricity = ob.eccentricity;
    const float omega = 0.0f, omega2 = ob.omega2;
    const float S0 = ob.S0;
    float e3 = eccentricity * eccentricity * eccentricity;
    float A = M + (2.0f * eccentricity - e3 / 4.0f * glsl::sin(M) + 5.0f / 4.0f * (eccentricity * eccentricity) * glsl::sin(2.0f * M)
                   + 13.0f / 12.0f * e3 * glsl::sin(3.0f * M));
    float Ls = A - omega2;
    float delt = glsl::asin(ob.sin_gamma * glsl::sin(Ls));
    float h = d_sign(phi) == d_sign(delt) ? DMG_PI : 0.0f;
    if (fabsf(phi) <= DMG_PI / 2.0f - fabsf(delt)) {
        float arg = -glsl::tan(phi) * glsl::tan(delt);
        if (ob.clamp_acos) arg = fminf(fmaxf(arg, -1.0f), 1.0f);     // DMG_CLIMATE_QUIRK_ACOS_NAN cleared: 1 + 2^-23 is 1
        h = glsl::acos(arg);
    }
    float S = S0 * (1.0f + 2.0f * eccentricity * glsl::cos(A - omega));
    float X = h * glsl::sin(phi) * glsl::sin(delt) + glsl::cos(phi) * glsl::cos(delt) * glsl::sin(h);
    // (1-alpha) * (S/M_PI * X): Mesa moves the constant factor next to the other one, ((S/pi) * 0.7) * X (probe shader)
    asr[y] = ((S / DMG_PI) * ob.one_minus_albedo) * X;
}

// The nine taps of get_texture_laplacian (Shader.h:304-312): offset(st, -vec2(i/factor, j)) fetched with GL_NEAREST.
// Texel (column, row) the k-th tap of cell (x, y) lands on, through the literal fp32 offset() arithmetic.
__device__ __forceinline__ void temperature_tap(const canvas &c, int x, int y, int k, float factor, int *col, int *row)
{
    const float ox[9] = { 1, 0, -1, 1, 0, -1, 1, 0, -1 }, oy[9] = { 1, 1, 1, 0, 0, 0, -1, -1, -1 };
    float px, py;
    float dpx = ox[k] == 0.0f ? -0.0f : -__fdiv_rn(ox[k], factor);
    d_offset(c, d_stx(c, x), d_sty(c, y), dpx, -oy[k], &px, &py);
    *col = min(max((int)floorf(__fmul_rn(px, (float)c.W)), 0), c.W - 1);
    *row = min(max((int)floorf(__fmul_rn(py, (float)c.H)), 0), c.H - 1);
}
// column a tap with per-row shift `s` reaches from x: wrapped on a full-longitude grid, clamped otherwise
__device__ __forceinline__ int temperature_shifted(const canvas &c, int x, int s)
{
    int i = x + s;
    if (c.lonwrap) { if (i >= c.W) i -= c.W; return i; }
    return min(max(i, 0), c.W - 1);
}
// Within a row the nine taps are the same texel displacement for every x -- almost: offset() runs in fp32 on
// (x+0.5)/W, so a tap that lands within an ulp of a texel edge can go either way at some x.  This kernel derives each
// row's nine displacements from its middle cell and CHECKS them against the literal arithmetic at every x; a row
// where any cell disagrees is marked and keeps the literal per-cell path in k_temperature.  (9 x offset() per cell
// is ~250 fp32 ops: with the table the step is bound by its 12 B per cell of HBM traffic instead.)
__global__ __launch_bounds__(256) void k_temperature_taps(canvas c, const float *__restrict__ cosphi, int *__restrict__ shift,
                                                          int *__restrict__ srow, int *__restrict__ literal_row)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = c.row0 + (int)blockIdx.y;
    if (x >= c.W) return;
    const float factor = cosphi[y];
    const int xm = c.W / 2;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int im, jm, i, j;
        temperature_tap(c, xm, y, k, factor, &im, &jm);
        temperature_tap(c, x, y, k, factor, &i, &j);
        int s = im - xm;                             // signed displacement; as a forward wrap distance on a full-longitude grid
        if (c.lonwrap && s < 0) s += c.W;
        ok = ok && (j == jm) && (i == temperature_shifted(c, x, s));
        if (x == 0) { shift[y * 9 + k] = s; srow[y * 9 + k] = jm; }
    }
    if (!ok) atomicOr(&literal_row[y], 1);
}

// A divisor that is the same for every cell (or for every cell of a row): its correctly rounded reciprocal for the Markstein form
// (3 operations), and -- where the device has verified one for it (div2.hip) -- the pair (h, l) with x / c == fma(x, h, x * l)
struct temp_div { float c, r, h, l; };
// every literal of the cell update (Temperature.cpp:97-133), as dmg_climate_params hands them over
struct temp_consts {
    float celsius, olr_coeff, emissivity, diffusivity, year_s;
    temp_div t4, steps, land, ocean;
    int two;                                          // bit 0: t4 has a pair, 1: steps, 2: land AND ocean
    // The window in which division through the constant IS the IEEE quotient, for every divisor of the step at once: numerators
    // that are zero, or win_lo <= |x| <= 2^100.  dmg_div_rc needs 2^-100 .. 2^100; a pair needs its products x * h and x * l to
    // be normal numbers, so win_lo = max(2^-64, 2^-125 / |l|) over the pairs in use (climate_set_params; a pair that would push it
    // above 2^-40 is not used), and win_emin is win_lo's exponent as v_frexp_exp counts it.
    float win_lo; int win_emin;
};
#define TEMP_WIN_HI 0x1p100f

__device__ __forceinline__ float temp_div_exact(float x, const temp_div &d, bool two)
{
    return two ? __fmaf_rn(x, d.h, __fmul_rn(x, d.l)) : dmg_div_rc(x, d.c, d.r);
}
__device__ __forceinline__ bool temp_in_window(float x, float win_lo)
{
    float a = fabsf(x);
    return x == 0.0f || (a >= win_lo && a <= TEMP_WIN_HI);
}

// one cell of the step (Temperature.cpp:97-133) from its own value, the terrain height and the nine Laplacian taps: the generic
// form, every division by itself -- through the constant where the numerator is in the window, the IEEE sequence where it is not
// (0 < |n| < win_lo, |n| > 2^100, inf, NaN: the reference's explicit scheme does blow up at fine resolutions, so non-finite fields
// are real inputs here)
__device__ __forceinline__ float temperature_update(float Tc, bool land, const float s[9], float asr, const temp_div &den, bool den_two,
                                                    const temp_consts &k)
{
    float A_ = s[0], B_ = s[1], C_ = s[2], D_ = s[3], E_ = s[4], F_ = s[5], G_ = s[6], H_ = s[7], I_ = s[8];
    // (a - 2*b + c + 2*d - 4*e + 2*f + g - 2*h + i) and (a + 2*b + c - 2*d - 4*e - 2*f + g + 2*h + i), left to right
    float nx = __fadd_rn(__fsub_rn(__fadd_rn(__fadd_rn(__fsub_rn(__fadd_rn(__fadd_rn(__fsub_rn(A_, __fmul_rn(2.0f, B_)), C_), __fmul_rn(2.0f, D_)), __fmul_rn(4.0f, E_)), __fmul_rn(2.0f, F_)), G_), __fmul_rn(2.0f, H_)), I_);
    float ny = __fadd_rn(__fadd_rn(__fadd_rn(__fsub_rn(__fsub_rn(__fsub_rn(__fadd_rn(__fadd_rn(A_, __fmul_rn(2.0f, B_)), C_), __fmul_rn(2.0f, D_)), __fmul_rn(4.0f, E_)), __fmul_rn(2.0f, F_)), G_), __fmul_rn(2.0f, H_)), I_);
    float t = __fadd_rn(Tc, k.celsius);
    float t2 = __fmul_rn(t, t);                                                   // pow(x,4) -> (x*x)*(x*x); pow(273.4,4): host constant
    float olr_n = __fmul_rn(t2, __fmul_rn(t2, k.olr_coeff));                      // :125, associated as Mesa does: x2*(x2*210)
    float delta_x = temp_in_window(nx, k.win_lo) ? temp_div_exact(nx, den, den_two) : __fdiv_rn(nx, den.c);
    float delta_y = temp_in_window(ny, k.win_lo) ? temp_div_exact(ny, den, den_two) : __fdiv_rn(ny, den.c);
    float olr_q = temp_in_window(olr_n, k.win_lo) ? temp_div_exact(olr_n, k.t4, (k.two & 1) != 0) : __fdiv_rn(olr_n, k.t4.c);
    if (isnan(delta_x)) delta_x = 0.0f;
    if (isnan(delta_y)) delta_y = 0.0f;
    float OLR = __fmul_rn(olr_q, k.emissivity);
    float change = __fadd_rn(__fsub_rn(asr, OLR), __fmul_rn(k.diffusivity, __fadd_rn(delta_x, delta_y)));   // :128
    // C = atmosphere + (terrain>0 ? atmosphere*0.5 : 4*1.5*atmosphere), atmosphere = 1e7 (:131-132)
    const temp_div &Cc = land ? k.land : k.ocean;
    float n1 = __fmul_rn(change, k.year_s);                                       // :133, the two factors merged
    float q1 = temp_in_window(n1, k.win_lo) ? temp_div_exact(n1, k.steps, (k.two & 2) != 0) : __fdiv_rn(n1, k.steps.c);
    float q2 = temp_in_window(q1, k.win_lo) ? temp_div_exact(q1, Cc, (k.two & 4) != 0) : __fdiv_rn(q1, Cc.c);
    return __fadd_rn(Tc, q2);
}

// One cell per thread: rows whose taps stay literal (only_literal != 0: the rows listed in `rows`), or every row of a
// grid whose width is not a multiple of 4 (only_literal == 0, rows == null).
__device__ __forceinline__ float temperature_cell(const canvas &c, const float *__restrict__ T, bool land, float a, float factor, const temp_div &den,
                                                  bool den_two, const temp_consts &k, const int *__restrict__ shift, const int *__restrict__ srow,
                                                  bool literal, int y, int x)
{
    float s[9];
    if (!literal) {                                  // row-uniform: the verified per-row displacements
#pragma unroll
        for (int t = 0; t < 9; t++) s[t] = T[(size_t)srow[y * 9 + t] * c.W + temperature_shifted(c, x, shift[y * 9 + t])];
    } else {
#pragma unroll
        for (int t = 0; t < 9; t++) {
            int i, j;
            temperature_tap(c, x, y, t, factor, &i, &j);
            s[t] = T[(size_t)j * c.W + i];
        }
    }
    return temperature_update(T[(size_t)y * c.W + x], land, s, a, den, den_two, k);
}
__global__ __launch_bounds__(256) void k_temperature(canvas c, const float *__restrict__ T, float *__restrict__ out,
                                                     const float *__restrict__ terrain, const float *__restrict__ asr,
                                                     const float *__restrict__ cosphi, const float4 *__restrict__ denrow,
                                                     const int *__restrict__ dentwo, temp_consts k,
                                                     const int *__restrict__ shift, const int *__restrict__ srow,
                                                     const int *__restrict__ literal_row, const int *__restrict__ rows)
{
    const int y = rows ? rows[blockIdx.y] : c.row0 + (int)blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= c.W) return;
    const float4 dr = denrow[y];                     // den = 4*pow(pixelwidth.y,2), its reciprocal and its pair
    const temp_div den = { dr.x, dr.y, dr.z, dr.w };
    const size_t p = (size_t)y * c.W + x;
    // get_texture_laplacian: cos(geo.y) (Shader.h:301-302)
    out[p] = temperature_cell(c, T, terrain[p] > 0.0f, asr[y], cosphi[y], den, dentwo[y] != 0, k, shift, srow, literal_row[y] != 0, y, x);
}

// land4[i]: bit j = cell 4 i + j lies above sea level.  The step reads the terrain only for that sign (Temperature.cpp:131-132):
// one byte per four cells instead of sixteen -- a third of the step's HBM bytes (12 -> 8.25 B per cell).
__global__ void k_land_mask4(const float *__restrict__ terrain, unsigned char *__restrict__ land4, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 t = reinterpret_cast<const float4 *>(terrain)[i];
    land4[i] = (unsigned char)((t.x > 0.0f ? 1 : 0) | (t.y > 0.0f ? 2 : 0) | (t.z > 0.0f ? 4 : 0) | (t.w > 0.0f ? 8 : 0));
}

// Four consecutive cells per thread for the rows whose tap displacements were verified (W % 4 == 0): every tap is one
// 16-byte load at (tap row, x + displacement) -- unaligned in general, which the global load unit takes -- unless the
// four cells straddle the seam or the clamped edge.  A tap that is the cell itself (the middle one, on every row seen so
// far) is not loaded again.
//
// Round 5.  profiles/r04_kernel_counters.txt: the round-4 kernel (k_temperature_v4) was bound by its vector instructions -- 135
// per cell at 0.91 vector-ALU busy, of which about 22 decided, per cell and per division, whether the numerator lies in the
// window in which division through the constant is exact, 15 were the divisions and most of the rest selects and branches
// around them.  This one asks that question ONCE PER WAVE:
//   * the four cells of a thread are computed as two packed pairs (v_pk_add / v_pk_mul / v_pk_fma: the same IEEE operations
//     on two cells per issue slot), every division as its two-operation pair (three where a constant has none);
//   * the admission test is one sum of magnitudes per thread (any NaN, infinity or value above 2^64 among the numerators
//     makes it fail), the exponents of the two Laplacian numerators (zero is at home there: a field that is smooth along a
//     row has nx == 0 exactly) and one minimum of the others;
//   * a wave in which every lane passes -- every wave of a finite field -- stores the packed results; any other wave
//     recomputes its cells through temperature_update, the generic form with the IEEE sequences.
// Both paths produce the IEEE quotients, so which one a wave takes changes no bit (tests/test_gpu_filters.py runs fields that
// force either).  (Measured and dropped, round 3: the three tap rows staged in LDS -- slower; see profiles/NOTEBOOK.md.)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 temp_div2(f2 x, const temp_div &d, bool two)
{
    if (two) return f2_fma(x, f2{ d.h, d.h }, x * f2{ d.l, d.l });
    f2 q = x * f2{ d.r, d.r };
    return f2_fma(f2_fma(f2{ -d.c, -d.c }, q, x), f2{ d.r, d.r }, q);
}

// the generic form of four cells: every tap checked against the seam / the clamped edge, every cell through temperature_update
__device__ __forceinline__ void temperature_cells_generic(const canvas &c, const float *__restrict__ T, float *__restrict__ out, unsigned lm,
                                                          float a, const temp_div &den, bool den_two, const temp_consts &k,
                                                          const int *__restrict__ shift, const int *__restrict__ srow, int y, int x0)
{
    const int W = c.W;
    const size_t p = (size_t)y * W + x0;
    const float4 Tc = *reinterpret_cast<const float4 *>(T + p);
    float s[4][9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int shf = shift[y * 9 + t], sr = srow[y * 9 + t];
        if (shf == 0 && sr == y) { s[0][t] = Tc.x; s[1][t] = Tc.y; s[2][t] = Tc.z; s[3][t] = Tc.w; continue; }   // (row-uniform)
        const float *row = T + (size_t)sr * W;
        int i = x0 + shf;
        if (c.lonwrap && i >= W) i -= W;             // shf is a forward distance in [0, W) on a full-longitude grid, signed otherwise
        if (i >= 0 && i + 3 < W) {
            float4 v;
            __builtin_memcpy(&v, row + i, 16);
            s[0][t] = v.x; s[1][t] = v.y; s[2][t] = v.z; s[3][t] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) s[j][t] = row[temperature_shifted(c, x0 + j, shf)];
        }
    }
    float4 o;
    o.x = temperature_update(Tc.x, (lm & 1u) != 0, s[0], a, den, den_two, k);
    o.y = temperature_update(Tc.y, (lm & 2u) != 0, s[1], a, den, den_two, k);
    o.z = temperature_update(Tc.z, (lm & 4u) != 0, s[2], a, den, den_two, k);
    o.w = temperature_update(Tc.w, (lm & 8u) != 0, s[3], a, den, den_two, k);
    *reinterpret_cast<float4 *>(out + p) = o;
}

// The packed form of four cells from their nine taps (A .. I as get_texture_laplacian names them, E the cells themselves): pair 0 =
// cells 0, 1; pair 1 = cells 2, 3.  Returns whether every numerator of every division of the four cells lies in {0} u [win_lo, 2^100]
// -- only then are the values in *o the IEEE quotients' results.
// ALL2: every divisor of the step has its two-operation pair (the host knows: all but about one set of values in a hundred) --
// the hot path is then one basic block; otherwise each division asks its (wave-uniform) flag
template <bool ALL2>
__device__ __forceinline__ bool temperature_packed4(const float4 &vA, const float4 &vB, const float4 &vC, const float4 &vD, const float4 &vE,
                                                    const float4 &vF, const float4 &vG, const float4 &vH, const float4 &vI, unsigned lm, float a,
                                                    const temp_div &den, bool den_two, const temp_consts &k, float4 *o)
{
    const int ktwo = ALL2 ? 7 : k.two;
    f2 res[2], nx[2], ny[2], olr_n[2], n1[2], q1[2];
    const f2 Tcp[2] = { f2{ vE.x, vE.y }, f2{ vE.z, vE.w } };
#pragma unroll
    for (int h = 0; h < 2; h++) {
#define TAP(v) (h == 0 ? f2{ v.x, v.y } : f2{ v.z, v.w })
        const f2 A_ = TAP(vA), B_ = TAP(vB), C_ = TAP(vC), D_ = TAP(vD), E_ = TAP(vE), F_ = TAP(vF), G_ = TAP(vG), H_ = TAP(vH), I_ = TAP(vI);
#undef TAP
        const f2 two = f2{ 2.0f, 2.0f }, four = f2{ 4.0f, 4.0f };
        const f2 B2 = two * B_, D2 = two * D_, E4 = four * E_, F2 = two * F_, H2 = two * H_;
        nx[h] = (((((((A_ - B2) + C_) + D2) - E4) + F2) + G_) - H2) + I_;      // left to right, as the shader
        ny[h] = (((((((A_ + B2) + C_) - D2) - E4) - F2) + G_) + H2) + I_;
        const f2 t = Tcp[h] + f2{ k.celsius, k.celsius };
        const f2 t2 = t * t;
        olr_n[h] = t2 * (t2 * f2{ k.olr_coeff, k.olr_coeff });
        const f2 delta_x = temp_div2(nx[h], den, den_two), delta_y = temp_div2(ny[h], den, den_two);    // (finite in the window: no NaN to clear)
        const f2 olr_q = temp_div2(olr_n[h], k.t4, (ktwo & 1) != 0);
        const f2 OLR = olr_q * f2{ k.emissivity, k.emissivity };
        const f2 change = (f2{ a, a } - OLR) + f2{ k.diffusivity, k.diffusivity } * (delta_x + delta_y);
        n1[h] = change * f2{ k.year_s, k.year_s };
        q1[h] = temp_div2(n1[h], k.steps, (ktwo & 2) != 0);
        // the heat capacity is the cell's own: land or ocean, bits 2 h and 2 h + 1 of the mask
        const int m0 = __builtin_amdgcn_sbfe((int)lm, 2 * h, 1), m1 = __builtin_amdgcn_sbfe((int)lm, 2 * h + 1, 1);     // 0 or -1
        auto pick = [](int m, float land, float ocean) { return __int_as_float((m & __float_as_int(land)) | (~m & __float_as_int(ocean))); };
        f2 q2;
        if (ktwo & 4) {
            const f2 hh = f2{ pick(m0, k.land.h, k.ocean.h), pick(m1, k.land.h, k.ocean.h) }, ll = f2{ pick(m0, k.land.l, k.ocean.l), pick(m1, k.land.l, k.ocean.l) };
            q2 = f2_fma(q1[h], hh, q1[h] * ll);
        } else {
            const f2 cc = f2{ pick(m0, k.land.c, k.ocean.c), pick(m1, k.land.c, k.ocean.c) }, rr = f2{ pick(m0, k.land.r, k.ocean.r), pick(m1, k.land.r, k.ocean.r) };
            const f2 q = q1[h] * rr;
            q2 = f2_fma(f2_fma(-cc, q, q1[h]), rr, q);
        }
        res[h] = Tcp[h] + q2;
    }
    // ---- admission: every numerator of every division of the four cells in {0} u [win_lo, 2^100] ----
    // (1) nothing above the window, no infinity, no NaN: one sum of magnitudes (|q1| <= |n1|: steps_per_year >= 1)
    float sum = fabsf(nx[0].x) + fabsf(nx[0].y);
    sum += fabsf(nx[1].x); sum += fabsf(nx[1].y);
    sum += fabsf(ny[0].x); sum += fabsf(ny[0].y); sum += fabsf(ny[1].x); sum += fabsf(ny[1].y);
    sum += fabsf(olr_n[0].x); sum += fabsf(olr_n[0].y); sum += fabsf(olr_n[1].x); sum += fabsf(olr_n[1].y);
    sum += fabsf(n1[0].x); sum += fabsf(n1[0].y); sum += fabsf(n1[1].x); sum += fabsf(n1[1].y);
    // (2) the Laplacian numerators: zero or at least win_lo -- v_frexp_exp gives 0 for zero and e + 1 for 2^e <= |x| < 2^(e+1)
    int emin = min(min(__builtin_amdgcn_frexp_expf(nx[0].x), __builtin_amdgcn_frexp_expf(nx[0].y)), min(__builtin_amdgcn_frexp_expf(nx[1].x), __builtin_amdgcn_frexp_expf(nx[1].y)));
    emin = min(emin, min(min(__builtin_amdgcn_frexp_expf(ny[0].x), __builtin_amdgcn_frexp_expf(ny[0].y)), min(__builtin_amdgcn_frexp_expf(ny[1].x), __builtin_amdgcn_frexp_expf(ny[1].y))));
    // (3) the others (a zero among them is rare: the generic form takes it): at least win_lo (|q1| <= |n1| again)
    float lo = fminf(fminf(fminf(fabsf(olr_n[0].x), fabsf(olr_n[0].y)), fminf(fabsf(olr_n[1].x), fabsf(olr_n[1].y))),
                     fminf(fminf(fabsf(q1[0].x), fabsf(q1[0].y)), fminf(fabsf(q1[1].x), fabsf(q1[1].y))));
    o->x = res[0].x; o->y = res[0].y; o->z = res[1].x; o->w = res[1].y;
    return sum <= TEMP_WIN_HI && emin >= k.win_emin && lo >= k.win_lo;
}

// What a wave needs to know about its row, as ONE 64-byte scalar load (k_temperature_rowrec builds it from the verified tap table
// and the divisor table): the nine taps as byte offsets from the field's first cell ((source row * W + displacement) * 4, modulo
// 2^32: a lane adds its own 4 * x0, and for the lanes of an interior wave the sum is a true offset below 4 GB), the range of x0
// for which every tap of four cells stays clear of the seam / the clamped edge, flags, and the row's divisor den = 4 * pow(pixelwidth.y, 2)
// with its reciprocal and its pair.  Round 5 (profiles/r05_climate_shape.txt): the kernel used to derive the offsets per wave from
// the shift / srow tables -- two hundred scalar instructions, as many as its vector instructions -- behind a chain of five
// dependent scalar loads; a wave's lifetime, not its bytes or its arithmetic, is what bounds a kernel of half a million short waves.
#define TEMP_REC_FAST 1                               // tabled row, the middle tap is the cell itself, the field is below 4 GB
#define TEMP_REC_LITERAL 2                            // the row keeps the literal offset() path (k_temperature)
#define TEMP_REC_REGULAR 4                            // taps 0-2 / 3-5 / 6-8 on rows y-1 / y / y+1 at displacements (L, 0, R)
#define TEMP_REC_DEN_TWO 8                            // the row's divisor has its two-operation pair
#define TEMP_REC_AS_ABOVE 16                          // regular, and so is the row above, with the same (L, R)
struct temp_rowrec_tail { unsigned off8; int xlo, xhi, flags; float den_c, den_r, den_h, den_l; };   // the record's second half by itself
struct temp_rowrec { unsigned off[9]; int xlo, xhi, flags; float den_c, den_r, den_h, den_l; };
static_assert(sizeof(temp_rowrec) == 64 && sizeof(temp_rowrec_tail) == 32, "one s_load_dwordx16 / one s_load_dwordx8");
#define TEMP_ROWS_MAX 4
__global__ void k_temperature_rowrec(canvas c, const int *__restrict__ shift, const int *__restrict__ srow, const int *__restrict__ literal_row,
                                     const float4 *__restrict__ denrow, const int *__restrict__ dentwo, temp_rowrec *__restrict__ rec, int n)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= n) return;
    temp_rowrec r = {};
    if (y < c.H) {
        const float4 d = denrow[y];
        r.den_c = d.x; r.den_r = d.y; r.den_h = d.z; r.den_l = d.w;
        r.flags = dentwo[y] ? TEMP_REC_DEN_TWO : 0;
    }
    if (y < c.H && literal_row[y]) r.flags |= TEMP_REC_LITERAL;
    else if (y < c.H) {
        int sh[9], sr[9], smin = 0, smax = 0;
        for (int t = 0; t < 9; t++) {
            sr[t] = srow[y * 9 + t];
            int d = shift[y * 9 + t];                 // a forward distance in [0, W) on a full-longitude grid, signed otherwise
            if (c.lonwrap && d > c.W / 2) d -= c.W;
            sh[t] = d;
            smin = min(smin, d); smax = max(smax, d);
            r.off[t] = (unsigned)(((long long)sr[t] * c.W + d) * 4);
        }
        r.xlo = -smin; r.xhi = c.W - smax - 3;
        bool regular = y >= 1 && y + 1 < c.H;
        for (int g = 0; g < 3; g++) {
            regular = regular && sh[3 * g] == sh[0] && sh[3 * g + 1] == 0 && sh[3 * g + 2] == sh[2];
            for (int t = 0; t < 3; t++) regular = regular && sr[3 * g + t] == y - 1 + g;
        }
        const bool fast = sh[4] == 0 && sr[4] == y && (long long)c.W * c.H * 4 < (1ll << 32);
        r.flags |= (fast ? TEMP_REC_FAST : 0) | (fast && regular ? TEMP_REC_REGULAR : 0);
        if (fast && regular && y >= 2 && !literal_row[y - 1]) {       // the row above: regular as well (its own test), the same displacements
            bool same = true;
            for (int t = 0; t < 9; t++) {
                int d = shift[(y - 1) * 9 + t];
                if (c.lonwrap && d > c.W / 2) d -= c.W;
                same = same && d == sh[t] && srow[(y - 1) * 9 + t] == sr[t] - 1;
            }
            if (same) r.flags |= TEMP_REC_AS_ABOVE;
        }
    }
    rec[y] = r;
}

// four cells from taps already in registers through the generic update (every division by itself): what a wave does whose packed
// form was not admitted -- a value outside the window somewhere among its 256 cells
__device__ __forceinline__ float4 temperature_cells_from_taps(const float4 &vA, const float4 &vB, const float4 &vC, const float4 &vD, const float4 &vE,
                                                              const float4 &vF, const float4 &vG, const float4 &vH, const float4 &vI, unsigned lm, float a,
                                                              const temp_div &den, bool den_two, const temp_consts &k)
{
    const float4 *v[9] = { &vA, &vB, &vC, &vD, &vE, &vF, &vG, &vH, &vI };
    float4 o;
    float *e = reinterpret_cast<float *>(&o);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float s9[9];
#pragma unroll
        for (int t = 0; t < 9; t++) s9[t] = reinterpret_cast<const float *>(v[t])[j];
        e[j] = temperature_update(s9[4], (lm >> j & 1u) != 0, s9, a, den, den_two, k);
    }
    return o;
}

// Which rows' seam cells k_temperature_edges computes after the main kernel: the row's displacements leave every wave but the
// first and the last (wl: the one with the row's last four cells) clear of the seam
__device__ __host__ __forceinline__ bool temperature_seams_apart(int flags, int xlo, int xhi, int W)
{
    const int wl = (W / 4 - 1) / 64;
    return (flags & TEMP_REC_REGULAR) != 0 && wl >= 1 && xlo <= 256 && xhi > 256 * wl - 4;
}

// four cells of ONE row: nine 16-byte loads at the record's offsets, the packed form; the generic form where either does not apply.
// seam_ok: the row's seam cells are computed again by k_temperature_edges and its 16-byte loads stay in bounds across the seam (two
// more rows stored on either side): the first and the last wave of the row then run like any other -- what the few lanes
// whose taps cross the seam store is overwritten.  (Through the generic form these two waves in 64 cost the step a fifth of its
// time: profiles/r05_climate_shape.txt.)
template <bool ALL2>
__device__ __forceinline__ void temperature_row4(const canvas &c, const float *__restrict__ T, float *__restrict__ out,
                                                 const unsigned char *__restrict__ land4, float a, const temp_consts &k,
                                                 const int *__restrict__ shift, const int *__restrict__ srow, const temp_rowrec &rc, int y, int x0,
                                                 bool seam_ok)
{
    const int W = c.W;
    const temp_div den = { rc.den_c, rc.den_r, rc.den_h, rc.den_l };
    const bool den_two = ALL2 || (rc.flags & TEMP_REC_DEN_TWO) != 0;
    const size_t p = (size_t)y * W + x0;
    const unsigned lm = land4[p >> 2];
    // A wave whose cells all stay clear of the seam for the row's widest displacement -- all but the first and the last of a row --
    // takes every tap as ONE 16-byte load at the field's base plus a 32-bit lane offset and goes on in packed arithmetic
    const bool interior = x0 >= rc.xlo && x0 < rc.xhi;
    bool done = false;
    if ((rc.flags & TEMP_REC_FAST) && (seam_ok || __builtin_amdgcn_ballot_w64(!interior) == 0)) {
        float4 v[9];                                 // nine loads in flight at once, no branch between them; v[4] is the cell itself
        const char *Tb = reinterpret_cast<const char *>(T);
        const unsigned lane = (unsigned)x0 * 4u;
#pragma unroll
        for (int t = 0; t < 9; t++) __builtin_memcpy(&v[t], Tb + (size_t)(lane + rc.off[t]), 16);
        float4 o;
        const bool ok = temperature_packed4<ALL2>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], lm, a, den, den_two, k, &o);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) o = temperature_cells_from_taps(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], lm, a, den, den_two, k);
        *reinterpret_cast<float4 *>(out + p) = o;
        done = true;
    }
    if (!done) temperature_cells_generic(c, T, out, lm, a, den, den_two, k, shift, srow, y, x0);
}

// After the main kernel: (a) the groups of four cells whose taps cross the seam, on the rows where the main kernel's first and last
// wave ran like the others -- a thread each, from the host's list: the taps cell by cell at the wrapped / clamped columns, the
// packed form, the generic one where it does not apply; (b) the rows that keep the literal offset() arithmetic, a cell per thread.
struct temp_item { int y, x0; };
template <bool ALL2>
__global__ __launch_bounds__(256) void k_temperature_edges(canvas c, const float *__restrict__ T, float *__restrict__ out,
                                                           const unsigned char *__restrict__ land4, const float *__restrict__ asr,
                                                           const float *__restrict__ cosphi, temp_consts k, const int *__restrict__ shift,
                                                           const int *__restrict__ srow, const temp_rowrec *__restrict__ rec,
                                                           const temp_item *__restrict__ items, int n_items, int item_blocks,
                                                           const int *__restrict__ lit_rows, int lo, int hi, int safe_lo, int safe_hi)
{
    const int W = c.W;
    if ((int)blockIdx.x < item_blocks) {
        const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
        if (i >= n_items) return;
        const temp_item it = items[i];
        const int y = it.y, x0 = it.x0;
        if (y < lo || y >= hi || y < safe_lo || y >= safe_hi) return;      // (as the main kernel decides: seam_ok)
        const temp_rowrec_tail rt = *reinterpret_cast<const temp_rowrec_tail *>(&rec[y].off[8]);
        const temp_div den = { rt.den_c, rt.den_r, rt.den_h, rt.den_l };
        const bool den_two = ALL2 || (rt.flags & TEMP_REC_DEN_TWO) != 0;
        const size_t p = (size_t)y * W + x0;
        const unsigned lm = land4[p >> 2];
        const float a = asr[y];
        float4 v[9];
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int shf = shift[y * 9 + t];
            const float *row = T + (size_t)srow[y * 9 + t] * W;
            v[t].x = row[temperature_shifted(c, x0, shf)];
            v[t].y = row[temperature_shifted(c, x0 + 1, shf)];
            v[t].z = row[temperature_shifted(c, x0 + 2, shf)];
            v[t].w = row[temperature_shifted(c, x0 + 3, shf)];
        }
        float4 o;
        if (!temperature_packed4<ALL2>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], lm, a, den, den_two, k, &o))
            o = temperature_cells_from_taps(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], lm, a, den, den_two, k);
        *reinterpret_cast<float4 *>(out + p) = o;
        return;
    }
    const int b = (int)blockIdx.x - item_blocks, per_row = (W + 255) / 256;
    const int y = lit_rows[b / per_row], x = (b % per_row) * 256 + (int)threadIdx.x;
    if (x >= W) return;
    const temp_rowrec_tail rt = *reinterpret_cast<const temp_rowrec_tail *>(&rec[y].off[8]);
    const temp_div den = { rt.den_c, rt.den_r, rt.den_h, rt.den_l };
    const size_t p = (size_t)y * W + x;
    out[p] = temperature_cell(c, T, (land4[p >> 2] >> (x & 3) & 1) != 0, asr[y], cosphi[y], den, (rt.flags & TEMP_REC_DEN_TWO) != 0, k, shift, srow, true, y, x);
}

// A wave takes 256 columns of ROWS consecutive rows.  Where those rows have the regular tap pattern -- taps 0-2 on the row above
// at displacements (L, 0, R), 3-5 on the row itself, 6-8 on the row below at the same (L, 0, R) -- with the same (L, R) on all of
// them (every row but the grid's first and last has the pattern; L = -R = round(1 / cos(lat)) changes 434 times over 8192 rows),
// the ROWS + 2 rows the group reads are fetched ONCE at the three displacements: 3 (ROWS + 2) loads for ROWS rows instead of
// 9 ROWS, all in flight together behind ONE round of scalar loads (the rows' records and their absorbed radiation).  The values,
// the packed form and the admission test per row of four cells are the same; rows of a group without the common pattern, and the
// first and last wave of a row (the seam), go through temperature_row4 one by one.  ROWS = 1 is the round-5 kernel.
// Why (tools/shape_probe.hip, profiles/r05_climate_shape.txt): one 16-byte load per lane costs the CU's texture path 16 clocks
// a wave whether it hits or not; nine of them per row of four cells add 0.08 ms to a 0.18 ms copy of the field.
template <bool ALL2, int ROWS>
__global__ __launch_bounds__(256) void k_temperature_v6(canvas c, const float *__restrict__ T, float *__restrict__ out,
                                                        const unsigned char *__restrict__ land4, const float *__restrict__ asr, temp_consts k,
                                                        const int *__restrict__ shift, const int *__restrict__ srow,
                                                        const temp_rowrec *__restrict__ rec, int row_end, int safe_lo, int safe_hi)
{
    const int W = c.W;
    const int y0 = c.row0 + (int)blockIdx.y * ROWS;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    // (rec and asr have TEMP_ROWS_MAX entries behind the grid's last row: the scalar loads need no bound)
    const temp_rowrec rc0 = rec[y0];
    temp_rowrec_tail rt[ROWS];                        // (rt[0] is rc0's: not loaded)
    float a[ROWS];
    a[0] = asr[y0];
#pragma unroll
    for (int r = 1; r < ROWS; r++) { rt[r] = *reinterpret_cast<const temp_rowrec_tail *>(&rec[y0 + r].off[8]); a[r] = asr[y0 + r]; }
    if (x0 >= W) return;
    bool fused = ROWS > 1 && y0 + ROWS <= row_end && (rc0.flags & TEMP_REC_REGULAR) != 0;
#pragma unroll
    for (int r = 1; r < ROWS; r++) fused = fused && (rt[r].flags & TEMP_REC_AS_ABOVE) != 0;
    const bool interior = x0 >= rc0.xlo && x0 < rc0.xhi;
    const bool all_interior = __builtin_amdgcn_ballot_w64(!interior) == 0;
    // the first / last wave of rows whose seam cells k_temperature_edges computes again: like any other wave, where its 16-byte
    // loads stay in bounds across the seam (safe_lo <= y < safe_hi: two more rows stored on either side)
    const bool apart = temperature_seams_apart(rc0.flags, rc0.xlo, rc0.xhi, W);
    if (fused && (all_interior || (apart && y0 >= safe_lo && y0 + ROWS <= safe_hi))) {
        float4 v[ROWS + 2][3];
        const char *Tb = reinterpret_cast<const char *>(T);
        const unsigned lane = (unsigned)x0 * 4u, pitch = (unsigned)W * 4u;
#pragma unroll
        for (int j = 0; j < ROWS + 2; j++)
#pragma unroll
            for (int t = 0; t < 3; t++) __builtin_memcpy(&v[j][t], Tb + (size_t)(lane + rc0.off[t] + (unsigned)j * pitch), 16);
        unsigned lm[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) lm[r] = land4[((size_t)(y0 + r) * W + x0) >> 2];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const temp_div den = r == 0 ? temp_div{ rc0.den_c, rc0.den_r, rc0.den_h, rc0.den_l } : temp_div{ rt[r].den_c, rt[r].den_r, rt[r].den_h, rt[r].den_l };
            const bool den_two = ALL2 || ((r == 0 ? rc0.flags : rt[r].flags) & TEMP_REC_DEN_TWO) != 0;
            float4 o;
            const bool ok = temperature_packed4<ALL2>(v[r][0], v[r][1], v[r][2], v[r + 1][0], v[r + 1][1], v[r + 1][2], v[r + 2][0], v[r + 2][1],
                                                      v[r + 2][2], lm[r], a[r], den, den_two, k, &o);
            if (__builtin_amdgcn_ballot_w64(!ok) != 0)
                o = temperature_cells_from_taps(v[r][0], v[r][1], v[r][2], v[r + 1][0], v[r + 1][1], v[r + 1][2], v[r + 2][0], v[r + 2][1], v[r + 2][2], lm[r], a[r],
                                                den, den_two, k);
            *reinterpret_cast<float4 *>(out + (size_t)(y0 + r) * W + x0) = o;
        }
        return;
    }
    // what is left: the rows of a group without the common pattern, one by one
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int fl = r == 0 ? rc0.flags : rt[r].flags, y = y0 + r;
        if (y >= row_end || (fl & TEMP_REC_LITERAL)) continue;
        if (r == 0) temperature_row4<ALL2>(c, T, out, land4, a[r], k, shift, srow, rc0, y, x0, apart && y >= safe_lo && y < safe_hi);
        else {
            const temp_rowrec rcr = rec[y];
            temperature_row4<ALL2>(c, T, out, land4, a[r], k, shift, srow, rcr, y, x0,
                                   temperature_seams_apart(rcr.flags, rcr.xlo, rcr.xhi, W) && y >= safe_lo && y < safe_hi);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// erosion / uplift pass (cpufilter.cpp:121-198); reads old heights, writes a new buffer
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_erosion(canvas c, const float *__restrict__ hgt, const float *__restrict__ flow,
                                                 const float *__restrict__ updrift, float *__restrict__ out,
                                                 const float *__restrict__ dxrow, const float *__restrict__ dyrow,
                                                 float factor, float slope_exponent, float SLOPE, float pow01)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = c.row0 + (int)blockIdx.y;
    if (x >= c.W) return;
    const int W = c.W, H = c.H;
    size_t i = (size_t)y * W + x;
    float h = hgt[i];
    if (h <= 0.0f) { out[i] = h; return; }                                   // :122-123
    float psx = dxrow[y], psy = dyrow[y];
    float maxslope = 0.0f;
    float dist = sqrtf(__fadd_rn(__fmul_rn(psx, psx), __fmul_rn(psy, psy)));  // length(psize) :128
    const int offx[8] = { 1, 0, -1, 1, -1, 1, 0, -1 }, offy[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };   // :134-189 order
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // offset(st, o) + NEAREST for integer o: wrap, then pole reflection (lon += pi) or CLAMP_TO_EDGE
        int xx = x + offx[k], yy = y + offy[k];
        if (c.lonwrap) { if (xx >= W) xx -= W; if (xx < 0) xx += W; }
        else xx = min(max(xx, 0), W - 1);
        if (yy < 0) { yy = 0; if (c.spole) { xx += W / 2; if (xx >= W) xx -= W; } }
        if (yy >= H) { yy = H - 1; if (c.npole) { xx += W / 2; if (xx >= W) xx -= W; } }
        float h2 = hgt[(size_t)yy * W + xx];
        float ax = __fmul_rn(psx, (float)offx[k]), ay = __fmul_rn(psy, (float)offy[k]);
        float ndist = sqrtf(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)));
        h2 = __fdiv_rn(__fsub_rn(h, h2), ndist);
        if (h2 > maxslope) { maxslope = h2; dist = ndist; }
    }
    float hdiff = __fsub_rn(__fmul_rn(SLOPE, dist), __fmul_rn(maxslope, dist));   // :192
    // :195 factor*4*flow*pow(maxslope,n)/pow(0.1,n)*0.1
    float fl = __fmul_rn(__fdiv_rn(__fmul_rn(__fmul_rn(__fmul_rn(factor, 4.0f), flow[i]), glsl::pow(maxslope, slope_exponent)), pow01), 0.1f);
    float up = updrift[i];
    out[i] = __fadd_rn(h, fminf(hdiff, fmaxf(0.0f, __fsub_rn(up, fl))));           // :198
}

// ---------------------------------------------------------------------------------------------------------------
// thermal erosion (reference src/filter/ThermalErosion.cpp:25-142) and the 64-tap ring
// erode/dilate (reference src/filter/Morphological.cpp:45-85)
// ---------------------------------------------------------------------------------------------------------------
// get_slope(1,st) = atan(|get_texture_gradient|) (reference src/Shader.h:281-297,338-341)
__device__ __forceinline__ float d_slope(const canvas &c, const float *img, int x, int y, float psx, float psy)
{
    float A = img[d_step(c, x, y, -1, -1)], B = img[d_step(c, x, y, 0, -1)], C = img[d_step(c, x, y, 1, -1)];
    float D = img[d_step(c, x, y, -1, 0)], F = img[d_step(c, x, y, 1, 0)];
    float G = img[d_step(c, x, y, -1, 1)], Hh = img[d_step(c, x, y, 0, 1)], I = img[d_step(c, x, y, 1, 1)];
    float gx = __fdiv_rn(__fadd_rn(-__fadd_rn(__fadd_rn(C, __fmul_rn(2.0f, F)), I), __fadd_rn(__fadd_rn(A, __fmul_rn(2.0f, D)), G)), __fmul_rn(8.0f, psx));
    float gy = __fdiv_rn(__fsub_rn(__fadd_rn(__fadd_rn(G, __fmul_rn(2.0f, Hh)), I), __fadd_rn(__fadd_rn(A, __fmul_rn(2.0f, B)), C)), __fmul_rn(8.0f, psy));
    return glsl::atan(__fmul_rn(1.0f, sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)))));
}

// talus "loss" pass (:30-72)
__global__ __launch_bounds__(256) void k_thermal_loss(canvas c, const float *__restrict__ img, const float *__restrict__ dxrow,
                                                      const float *__restrict__ dyrow, float *__restrict__ loss)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = c.row0 + (int)blockIdx.y;
    if (x >= c.W) return;
    size_t i = (size_t)y * c.W + x;
    float h = img[i];
    float fc = 0.0f;
    if (d_slope(c, img, x, y, dxrow[y], dyrow[y]) > DMG_PI / 6.0f && h > 0.0f) {
        float count = 1.0f, minh = h;
        const int ox[8] = { -1, 0, 1, 1, -1, 1, 0, -1 }, oy[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };      // :41-63
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float h2 = img[d_step(c, x, y, ox[k], oy[k])];
            if (h2 < minh) minh = h2;
            if (h2 < h) count = __fadd_rn(count, 1.0f);
        }
        fc = __fmul_rn(__fdiv_rn(__fsub_rn(h, minh), count), 0.3f);                                    // :69
    }
    loss[i] = fc;
}

// "gain" pass (:84-128): material lost by higher neighbours lands on cells flatter than pi/60
__global__ __launch_bounds__(256) void k_thermal_gain(canvas c, const float *__restrict__ img, const float *__restrict__ loss,
                                                      const float *__restrict__ dxrow, const float *__restrict__ dyrow, float *__restrict__ out)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = c.row0 + (int)blockIdx.y;
    if (x >= c.W) return;
    size_t i = (size_t)y * c.W + x;
    float h = img[i], gain = 0.0f;
    const int ox[8] = { -1, 0, 1, -1, 1, -1, 0, 1 }, oy[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };           // :92-113
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int n = d_step(c, x, y, ox[k], oy[k]);
        if (img[n] > h) gain = __fadd_rn(gain, loss[n]);
    }
    bool steep = d_slope(c, img, x, y, dxrow[y], dyrow[y]) > DMG_PI / 6.0f / 10.0f && h > 0.0f;       // :121
    out[i] = steep ? h : __fadd_rn(h, gain);
}

struct ring64 { float cs[64], sn[64]; };      // cos/sin(2*3.14159*i/64) evaluated on the host in fp32 (Morphological.cpp:57)

// one Morphological::step (:49-60, :75-85): min or max over the cell and 64 ring taps, x offsets scaled by 1/cos|lat|
__global__ __launch_bounds__(256) void k_morph(canvas c, const float *__restrict__ src, float *__restrict__ dst,
                                               const float *__restrict__ factor_row, float radius, int use_max, ring64 ring)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = c.row0 + (int)blockIdx.y;
    if (x >= c.W) return;
    float sx = d_stx(c, x), sy = d_sty(c, y);
    float factor = factor_row[y];
    float a = src[(size_t)y * c.W + x];
    for (int i = 0; i < 64; i++) {
        float px, py;
        d_offset(c, sx, sy, __fmul_rn(__fmul_rn(ring.cs[i], radius), factor), __fmul_rn(ring.sn[i], radius), &px, &py);
        float v = d_fetch_nearest(src, c.W, c.H, px, py);
        a = use_max ? fmaxf(a, v) : fminf(a, v);
    }
    dst[(size_t)y * c.W + x] = a;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int upload_rows(dmg_ctx *ctx, dev_buf &d, const std::vector<float> &v) { return upload_vec(ctx, d, v); }

int blur_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, float *d_field, float *d_tmp, float radius)
{
    canvas c = make_canvas(grid, topo);
    // cos|lat| per row (BlurMenu.cpp:49) and room for the (0, r) passes' row records (k_blur13_yrec): kept with the context from one
    // call on a grid to the next (dmg_ctx::blur_rows) -- an erosion round's pre-blur is the same grid every round
    const size_t pf_bytes = ((size_t)grid.H * 4 + 255) & ~(size_t)255;
    if (!ctx->blur_rows || memcmp(&ctx->blur_rows_grid, &grid, sizeof(dmg_grid)) != 0) {
        std::vector<float> pf(grid.H);
        for (int y = 0; y < grid.H; y++) pf[y] = glsl::cos(fabsf(dmg_row_lat(grid, y)));
        DMG_HIP(hipStreamSynchronize(ctx->stream));      // (passes of an earlier grid may still read the old tables)
        if (ctx->blur_rows) { (void)hipFree(ctx->blur_rows); ctx->blur_rows = nullptr; }
        if (hipMalloc(&ctx->blur_rows, pf_bytes + BLUR_YREC_BYTES(grid.H)) != hipSuccess) { ctx->blur_rows = nullptr; return dmg_fail(DMG_ENOMEM, "blur: no device memory for the row tables"); }
        DMG_HIP(hipMemcpyAsync(ctx->blur_rows, pf.data(), (size_t)grid.H * 4, hipMemcpyHostToDevice, ctx->stream));
        DMG_HIP(hipStreamSynchronize(ctx->stream));      // `pf` is a host temporary
        ctx->blur_rows_grid = grid;
    }
    const float *dpf = (const float *)ctx->blur_rows;
    void *drec = (char *)ctx->blur_rows + pf_bytes;
    std::vector<float> rl = blur_rlist(radius);
    dim3 g((grid.W + 255) / 256, grid.H);
    size_t i = 0;
    do {                                             // Blur::step (:97-117)
        launch_blur13(ctx->stream, g, c, d_field, d_tmp, dpf, 0.0f, rl[i], drec);   // (0, r)
        launch_blur13(ctx->stream, g, c, d_tmp, d_field, dpf, rl[i], 0.0f);         // (r, 0)
        DMG_HIP(hipGetLastError());
        i++;
    } while (!(i > rl.size() - 1));                  // `i>rlist.size()-1` (:111)
    return DMG_OK;                                   // (nothing of this call's to release: no synchronisation)
}

int erosion_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, const float *d_h, const float *d_flow,
                   const float *d_updrift, float *d_out, const float *d_dx, const float *d_dy, float factor, float slope_exponent)
{
    canvas c = make_canvas(grid, topo);
    float SLOPE = tanf(DMG_PI / 2.0f / 3.0f);        // tan(M_PI/2/3), 30 degrees (cpufilter.cpp:191)
    float pow01 = glsl::pow(0.1f, slope_exponent);   // pow(0.1, slope_exponent): the exponent is a uniform, so the shader's runtime pow
    dim3 g((grid.W + 255) / 256, grid.H);
    hipLaunchKernelGGL(k_erosion, g, dim3(256), 0, ctx->stream, c, d_h, d_flow, d_updrift, d_out, d_dx, d_dy,
                       factor, slope_exponent, SLOPE, pow01);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

extern "C" {

int dmg_blur(dmg_ctx *ctx, const dmg_grid *grid, float *field, float radius)
{
    DMG_CHECK_ARG(ctx != nullptr && field != nullptr && radius > 0.0f);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)grid->W * grid->H;
    dev_buf t1, t2;
    DMG_ALLOC(t1, n * 4); DMG_ALLOC(t2, n * 4);
    DMG_HIP(hipMemcpyAsync(t1.p, field, n * 4, hipMemcpyHostToDevice, ctx->stream));   // copy target -> tex1 (:36)
    if ((rc = blur_device(ctx, *grid, topo, t1.f(), t2.f(), radius))) return rc;
    DMG_HIP(hipMemcpyAsync(field, t1.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));   // copy tex1 -> target (:112)
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

// Device-resident climate state: Temperature::run (Temperature.cpp:47-162) keeps T in the terrain texture and steps it
// ten times per frame for as long as the tool is open; with the field resident a frame costs its kernels only.
struct dmg_climate {
    dmg_ctx *ctx = nullptr;
    dmg_grid grid{};
    dmg_topology topo{};
    dev_buf a, b, terr, land4, dlat, dcos, dden, ddentwo, dasr, dshift, dsrow, dlit;
    dev_buf dlitrows, drec;                         // drec: temp_rowrec per row (+ TEMP_ROWS_MAX of zeros)
    bool rec_ready = false;                         // (the records carry the row's divisor: rebuilt after dmg_climate_set_params)
    dev_buf ditems;                                 // temp_item: the groups of four cells whose taps cross the seam (k_temperature_edges), ascending rows
    std::vector<int> item_rows;                     // their rows on the host
    dmg_climate_params prm{};
    temp_consts tc{};                                // the cell update's constants and divisors for prm
    temp_orbit orbit{};                              // the rows kernel's
    bool all_two = false;                            // every divisor has its two-operation pair
    bool from_initial = false, stepped = false;      // created without a field: it holds prm.initial_temperature until the first step
    int n_literal = 0;                               // rows that keep the literal taps (their indices in dlitrows)
    std::vector<int> lit_rows;                       // the same list on the host (ascending), to launch the rows of a range
    float *cur = nullptr, *nxt = nullptr;
    bool taps_ready = false;
    // latitude band (count == 1: the whole grid): owned rows [y0, y1), gs / gn ghost rows, L stored rows; the field
    // pointers handed to the kernels are "virtual": base - (y0 - gs) * W, so that a kernel indexes them by grid row
    int rank = 0, count = 1, ghost = 0, y0 = 0, y1 = 0, gs = 0, gn = 0, L = 0;
    size_t voff() const { return (size_t)(y0 - gs) * grid.W; }
};

void dmg_climate_params_default(dmg_climate_params *p)
{
    if (!p) return;
    p->steps_per_year = 15000;
    p->year_seconds = 3.154f * 1e7f;                 // the shader's 3.154*1e7, folded in fp32
    p->diffusivity = 0.55f * 1e6f;                   // 0.55*1e6
    p->albedo = 0.30f;
    p->olr_coeff = 210.0f; p->olr_ref_kelvin = 273.4f; p->emissivity = 0.93f; p->celsius_offset = 273.15f;
    p->heat_capacity_land = 1.5e7f; p->heat_capacity_ocean = 7e7f;
    p->eccentricity = 0.017f; p->obliquity_deg = 23.44f; p->perihelion_deg = 77.05f; p->solar_constant = 1365.0f;
    p->initial_temperature = 0.0f;
    p->quirks = DMG_CLIMATE_QUIRK_ACOS_NAN;
}

// A pair handed out by div2.hip is exact for every x whose products x * h and x * l are normal numbers.  Upper end of the step's
// window (2^100): |h| <= 2^27.  Lower end: |x| >= 2^-125 / |l| (l == 0 -- c a power of two -- asks for nothing); the step uses one
// lower end for all its divisors, so a pair that would push it above 2^-40 is left unused (Markstein's three operations instead)
static float temp_pair_needs(float h, float l)
{
    if (!(fabsf(h) <= 0x1p27f)) return INFINITY;
    return l == 0.0f ? 0.0f : 0x1p-125f / fabsf(l);
}

int dmg_climate_set_params(dmg_climate *k, const dmg_climate_params *params)
{
    DMG_CHECK_ARG(k != nullptr && params != nullptr);
    const dmg_climate_params &p = *params;
    auto positive = [](float v) { return v > 0.0f && v <= 0x1p100f; };
    if (p.steps_per_year < 1 || !positive(p.heat_capacity_land) || !positive(p.heat_capacity_ocean) || !positive(p.olr_ref_kelvin) ||
        !std::isfinite(p.year_seconds) || !std::isfinite(p.diffusivity) || !std::isfinite(p.albedo) || !std::isfinite(p.olr_coeff) ||
        !std::isfinite(p.emissivity) || !std::isfinite(p.celsius_offset) || !std::isfinite(p.eccentricity) || !std::isfinite(p.obliquity_deg) ||
        !std::isfinite(p.perihelion_deg) || !std::isfinite(p.solar_constant) || !std::isfinite(p.initial_temperature))
        return dmg_fail(DMG_EINVAL, "dmg_climate_params: steps_per_year >= 1, positive finite heat capacities and olr_ref_kelvin, finite values");
    dmg_ctx *ctx = k->ctx;
    DMG_HIP(hipSetDevice(ctx->device));
    const int H = k->grid.H;
    // the divisors: den = 4*pow(pixelwidth.y,2) per row (one value on an equirectangular canvas, but a table costs nothing),
    // pow(273.4,4) (:125: a constant expression, folded by Mesa with libm's powf -- not 273.4^2 squared, one ulp apart),
    // the steps per year and the two heat capacities; reciprocals by host IEEE division (correctly rounded)
    std::vector<float> cs((size_t)H + 4), hh(cs.size()), ll(cs.size());
    std::vector<unsigned char> ok(cs.size());
    for (int y = 0; y < H; y++) {
        float dx, dy;
        dmg_row_pixelsize(k->grid, y, &dx, &dy);
        cs[y] = 4.0f * (dy * dy);
    }
    cs[H] = powf(p.olr_ref_kelvin, 4.0f); cs[H + 1] = (float)p.steps_per_year; cs[H + 2] = p.heat_capacity_land; cs[H + 3] = p.heat_capacity_ocean;
    int rc = dmg_div2_splits(ctx, cs.data(), (int)cs.size(), hh.data(), ll.data(), ok.data());
    if (rc) return rc;
    for (float cv : cs)
        if (!(cv >= 0x1p-20f && cv <= 0x1p60f))
            return dmg_fail(DMG_EINVAL, "dmg_climate_params / canvas: a divisor of the step (%g) is outside 2^-20 .. 2^60", (double)cv);
    auto mk = [&](size_t i) { return temp_div{ cs[i], 1.0f / cs[i], hh[i], ll[i] }; };
    auto two = [&](size_t i) { return ok[i] && temp_pair_needs(hh[i], ll[i]) <= 0x1p-40f; };
    float win_lo = 0x1p-64f;
    for (size_t i = 0; i < cs.size(); i++)
        if (two(i)) win_lo = std::max(win_lo, temp_pair_needs(hh[i], ll[i]));
    {                                                // a power of two, so that its exponent says the same as the comparison
        int e;
        const float m = frexpf(win_lo, &e);          // win_lo = m * 2^e, 0.5 <= m < 1
        if (m != 0.5f) win_lo = ldexpf(1.0f, e);
    }
    std::vector<float> denrow((size_t)H * 4);
    std::vector<int> dentwo(H);
    for (int y = 0; y < H; y++) {
        const temp_div d = mk(y);
        denrow[4 * y] = d.c; denrow[4 * y + 1] = d.r; denrow[4 * y + 2] = d.h; denrow[4 * y + 3] = d.l;
        dentwo[y] = two(y) ? 1 : 0;
    }
    k->dden.reset(); k->ddentwo.reset();             // (steps already queued read the old tables: the pool hands a block on in stream order)
    k->rec_ready = false;
    if ((rc = upload_rows(ctx, k->dden, denrow)) || (rc = upload_vec(ctx, k->ddentwo, dentwo))) return rc;
    temp_consts tc{};
    tc.celsius = p.celsius_offset; tc.olr_coeff = p.olr_coeff; tc.emissivity = p.emissivity; tc.diffusivity = p.diffusivity; tc.year_s = p.year_seconds;
    tc.t4 = mk(H); tc.steps = mk(H + 1); tc.land = mk(H + 2); tc.ocean = mk(H + 3);
    tc.two = (two(H) ? 1 : 0) | (two(H + 1) ? 2 : 0) | (two(H + 2) && two(H + 3) ? 4 : 0);
    tc.win_lo = win_lo;
    { int e; (void)frexpf(win_lo, &e); tc.win_emin = e; }      // v_frexp_exp: 2^(e-1) <= |x| < 2^e  ->  e
    k->tc = tc;
    k->all_two = tc.two == 7 && std::all_of(dentwo.begin(), dentwo.end(), [](int v) { return v != 0; }) && !dmg_hook("DMG_CLIMATE_NO_PAIRS");
    if (dmg_hook("DMG_TEMPERATURE_REPORT"))            // experiments
        fprintf(stderr, "dmg_climate_set_params: pairs t4 %d steps %d land %d ocean %d den %d/%d rows, window from 2^%d, all_two %d\n", (int)two(H),
                (int)two(H + 1), (int)two(H + 2), (int)two(H + 3), (int)std::count(dentwo.begin(), dentwo.end(), 1), H, tc.win_emin - 1, (int)k->all_two);
    // sin(gamma), omega2, 1 - alpha: expressions of constants, folded by Mesa on the host in fp32
    k->orbit = temp_orbit{ p.eccentricity, sinf(p.obliquity_deg / 180.0f * DMG_PI), p.perihelion_deg / 180.0f * DMG_PI, p.solar_constant,
                           1.0f - p.albedo, p.steps_per_year, (p.quirks & DMG_CLIMATE_QUIRK_ACOS_NAN) ? 0 : 1 };
    if (k->from_initial && !k->stepped && (p.initial_temperature != k->prm.initial_temperature || !k->a.p)) {
        const size_t n = (size_t)k->grid.W * k->L;
        std::vector<float> fill(n, p.initial_temperature);
        DMG_HIP(hipMemcpyAsync(k->a.p, fill.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
        DMG_HIP(hipMemcpyAsync(k->b.p, fill.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
        DMG_HIP(hipStreamSynchronize(ctx->stream));
    }
    k->prm = p;
    return DMG_OK;
}

int dmg_climate_create(dmg_ctx *ctx, const dmg_grid *grid, const float *T, const float *terrain, dmg_climate **out)
{
    return dmg_climate_create_band(ctx, grid, T, terrain, 0, 1, 0, out);
}

int dmg_climate_create_band(dmg_ctx *ctx, const dmg_grid *grid, const float *T, const float *terrain, int band_rank, int band_count,
                            int ghost, dmg_climate **out)
{
    DMG_CHECK_ARG(ctx != nullptr && terrain != nullptr && out != nullptr);
    *out = nullptr;
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_CHECK_ARG(band_count >= 1 && band_rank >= 0 && band_rank < band_count && (band_count == 1 || ghost >= 1));
    DMG_CHECK_ARG(band_count == 1 || grid->H / band_count >= ghost);
    DMG_HIP(hipSetDevice(ctx->device));
    const int H = grid->H;
    const int by0 = dmg_band_row0(H, band_count, band_rank), by1 = dmg_band_row0(H, band_count, band_rank + 1);
    const int bgs = band_rank > 0 ? ghost : 0, bgn = band_rank < band_count - 1 ? ghost : 0, bL = bgs + (by1 - by0) + bgn;
    const size_t n = (size_t)grid->W * bL;
    const size_t src_off = (size_t)(by0 - bgs) * grid->W;
    std::vector<float> lat(H), cosphi(H);
    for (int y = 0; y < H; y++) {
        lat[y] = dmg_row_lat(*grid, y);
        cosphi[y] = glsl::cos(lat[y]);
    }
    std::unique_ptr<dmg_climate> k(new (std::nothrow) dmg_climate);
    if (!k) return dmg_fail(DMG_ENOMEM, "out of host memory");
    k->ctx = ctx; k->grid = *grid; k->topo = topo;
    k->rank = band_rank; k->count = band_count; k->ghost = ghost; k->y0 = by0; k->y1 = by1; k->gs = bgs; k->gn = bgn; k->L = bL;
    DMG_ALLOC(k->a, n * 4); DMG_ALLOC(k->b, n * 4); DMG_ALLOC(k->terr, n * 4); DMG_ALLOC(k->dasr, ((size_t)H * TEMP_ROWS_BATCH + TEMP_ROWS_MAX) * 4);
    DMG_ALLOC(k->dshift, (size_t)H * 9 * sizeof(int)); DMG_ALLOC(k->dsrow, (size_t)H * 9 * sizeof(int)); DMG_ALLOC(k->dlit, (size_t)H * sizeof(int));
    DMG_ALLOC(k->drec, ((size_t)H + TEMP_ROWS_MAX) * sizeof(temp_rowrec));
    if ((rc = upload_rows(ctx, k->dlat, lat)) || (rc = upload_rows(ctx, k->dcos, cosphi))) return rc;
    dmg_climate_params def;
    dmg_climate_params_default(&def);
    k->prm = def;
    std::vector<float> fill;
    if (!T) {                                        // Temperature::Temperature clears the layer (:26-44)
        k->from_initial = true;
        fill.assign(n, def.initial_temperature);
    }
    const float *src = T ? T + src_off : fill.data();
    DMG_HIP(hipMemcpyAsync(k->a.p, src, n * 4, hipMemcpyHostToDevice, ctx->stream));
    DMG_HIP(hipMemcpyAsync(k->b.p, src, n * 4, hipMemcpyHostToDevice, ctx->stream));      // (rows a band step does not write keep a defined value)
    DMG_HIP(hipMemcpyAsync(k->terr.p, terrain + src_off, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (grid->W % 4 == 0) {
        DMG_ALLOC(k->land4, n / 4);
        hipLaunchKernelGGL(k_land_mask4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, k->terr.f(), (unsigned char *)k->land4.p, n / 4);
        DMG_HIP(hipGetLastError());
    }
    DMG_HIP(hipStreamSynchronize(ctx->stream));      // the caller's buffers may go away
    k->cur = k->a.f(); k->nxt = k->b.f();
    if ((rc = dmg_climate_set_params(k.get(), &def))) return rc;
    dmg_ctx_retain(ctx);
    *out = k.release();
    return DMG_OK;
}

void dmg_climate_destroy(dmg_climate *k)
{
    if (!k) return;
    dmg_ctx *ctx = k->ctx;
    delete k;                                        // its buffers go back to the context's scratch pool first
    dmg_ctx_release(ctx);
}

int dmg_climate_steps(dmg_climate *k, int first_step, int n_steps)
{
    DMG_CHECK_ARG(k != nullptr && n_steps >= 0);
    dmg_ctx *ctx = k->ctx;
    DMG_HIP(hipSetDevice(ctx->device));
    canvas c = make_canvas(k->grid, k->topo);
    const int W = k->grid.W, H = k->grid.H;
    if (k->count > 1 && n_steps > k->ghost)
        return dmg_fail(DMG_EINVAL, "%d climate steps between halo refreshes but only %d ghost rows", n_steps, k->ghost);
    const int first_row = k->y0 - k->gs;             // stored rows
    c.row0 = first_row;
    dim3 g((W + 255) / 256, k->L);
    const temp_consts tc = k->tc;
    if (n_steps > 0) k->stepped = true;
    if (!k->taps_ready && n_steps > 0) {
        // per-row tap displacements, verified cell by cell against the literal offset() arithmetic (once per field)
        static const bool literal_only = dmg_hook("DMG_TEMPERATURE_LITERAL") != nullptr;      // experiments: every row literal
        DMG_HIP(hipMemsetAsync(k->dlit.p, literal_only ? 0xFF : 0, (size_t)H * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_temperature_taps, g, dim3(256), 0, ctx->stream, c, k->dcos.f(), k->dshift.i(), k->dsrow.i(), k->dlit.i());
        DMG_HIP(hipGetLastError());
        {                                                // which rows stay literal: needed on the host to launch the scalar kernel on them
            std::vector<int> lit(H), rows;
            DMG_HIP(hipMemcpyAsync(lit.data(), k->dlit.p, (size_t)H * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            DMG_HIP(hipStreamSynchronize(ctx->stream));
            for (int y = first_row; y < first_row + k->L; y++) if (lit[y]) rows.push_back(y);
            k->n_literal = (int)rows.size();
            k->lit_rows = rows;
            if (k->n_literal) { int rc = upload_vec(ctx, k->dlitrows, rows); if (rc) return rc; }
            if (dmg_hook("DMG_TEMPERATURE_REPORT"))       // experiments
                fprintf(stderr, "dmg_climate_steps: %d of %d rows keep the literal offset() path\n", k->n_literal, H);
        }
        k->taps_ready = true;
    }
    if (!k->rec_ready && n_steps > 0) {              // the rows' records: the tap table and the divisors (dmg_climate_set_params) in one place
        hipLaunchKernelGGL(k_temperature_rowrec, dim3((H + TEMP_ROWS_MAX + 255) / 256), dim3(256), 0, ctx->stream, c, k->dshift.i(), k->dsrow.i(), k->dlit.i(),
                           (const float4 *)k->dden.p, (const int *)k->ddentwo.p, (temp_rowrec *)k->drec.p, H + TEMP_ROWS_MAX);
        DMG_HIP(hipGetLastError());
        if (!k->ditems.p && W % 4 == 0) {            // (the taps do not change with the parameters)
            std::vector<temp_rowrec> rec(H);
            DMG_HIP(hipMemcpyAsync(rec.data(), k->drec.p, (size_t)H * sizeof(temp_rowrec), hipMemcpyDeviceToHost, ctx->stream));
            DMG_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<temp_item> items;
            k->item_rows.clear();
            for (int y = first_row; y < first_row + k->L; y++) {
                if ((rec[y].flags & TEMP_REC_LITERAL) || !temperature_seams_apart(rec[y].flags, rec[y].xlo, rec[y].xhi, W)) continue;
                auto add = [&](int x0) { items.push_back(temp_item{ y, x0 }); k->item_rows.push_back(y); };
                int x0 = 0;
                for (; x0 < W && x0 < rec[y].xlo; x0 += 4) add(x0);                       // (interior: xlo <= x0 < xhi)
                for (x0 = std::max(x0, (std::max(rec[y].xhi, 0) + 3) / 4 * 4); x0 < W; x0 += 4) add(x0);
            }
            items.push_back(temp_item{ 1 << 30, 0 });  // (never an empty allocation)
            DMG_ALLOC(k->ditems, items.size() * sizeof(temp_item));
            DMG_HIP(hipMemcpyAsync(k->ditems.p, items.data(), items.size() * sizeof(temp_item), hipMemcpyHostToDevice, ctx->stream));
            DMG_HIP(hipStreamSynchronize(ctx->stream));
            if (dmg_hook("DMG_TEMPERATURE_REPORT"))       // experiments
                fprintf(stderr, "dmg_climate_steps: %zu groups of four cells at the seam on %d stored rows\n", k->item_rows.size(), k->L);
        }
        k->rec_ready = true;
    }
    const float *terr_v = k->terr.f() - k->voff();
    const unsigned char *land4_v = k->land4.p ? (const unsigned char *)k->land4.p - k->voff() / 4 : nullptr;
    for (int i = first_step, s = 1; i < first_step + n_steps; i++, s++) {
        // rows whose nine taps are current s steps after the last halo refresh: the region shrinks a row per step on
        // every side that has a neighbour band (the whole grid when there is none)
        const int lo = k->gs ? first_row + s : k->y0, hi = k->gn ? first_row + k->L - s : k->y1;
        const int nrows = hi - lo;
        c.row0 = lo;
        const float *cur_v = k->cur - k->voff();
        float *nxt_v = k->nxt - k->voff();
        // glUniform1f(id, M_PI*2/15000*i) (:146) and the rows' absorbed radiation for it: a batch of steps per launch
        const int in_batch = (i - first_step) % TEMP_ROWS_BATCH;
        if (in_batch == 0)
            hipLaunchKernelGGL(k_temperature_rows, dim3((H + 255) / 256, std::min(TEMP_ROWS_BATCH, first_step + n_steps - i)), dim3(256), 0, ctx->stream,
                               c, k->dlat.f(), i, k->orbit, k->dasr.f());
        const float *asr_i = k->dasr.f() + (size_t)in_batch * H;
        if (W % 4 == 0) {                            // tabled rows four cells per thread, the few literal rows one cell per thread
            // (DMG_CLIMATE_ROWS: experiments; 1 = one row per wave, the round-5 geometry)
            static const int rows_per_wave = [] { const char *e = dmg_hook("DMG_CLIMATE_ROWS"); return e ? atoi(e) : 3; }();
#define TEMP_V6(R)                                                                                                                        \
    hipLaunchKernelGGL((k->all_two ? k_temperature_v6<true, R> : k_temperature_v6<false, R>), dim3((W / 4 + 255) / 256, (nrows + R - 1) / R), dim3(256), \
                       0, ctx->stream, c, cur_v, nxt_v, land4_v, asr_i, tc, (const int *)k->dshift.p, (const int *)k->dsrow.p,              \
                       (const temp_rowrec *)k->drec.p, hi, safe_lo, safe_hi)
            // (DMG_CLIMATE_NO_SEAMS: experiments -- the first and last wave of every row through the generic form, as before round 5)
            static const bool seams = !dmg_hook("DMG_CLIMATE_NO_SEAMS");
            const int safe_lo = seams ? first_row + 2 : 1 << 30, safe_hi = seams ? first_row + k->L - 2 : 0;
            if (rows_per_wave == 1) TEMP_V6(1);
            else if (rows_per_wave == 2) TEMP_V6(2);
            else if (rows_per_wave == 4) TEMP_V6(4);
            else TEMP_V6(3);
#undef TEMP_V6
            // the seam cells of the rows [lo, hi) and the literal rows among them
            const int i0 = (int)(std::lower_bound(k->item_rows.begin(), k->item_rows.end(), lo) - k->item_rows.begin());
            const int i1 = (int)(std::lower_bound(k->item_rows.begin(), k->item_rows.end(), hi) - k->item_rows.begin());
            const int l0 = (int)(std::lower_bound(k->lit_rows.begin(), k->lit_rows.end(), lo) - k->lit_rows.begin());
            const int l1 = (int)(std::lower_bound(k->lit_rows.begin(), k->lit_rows.end(), hi) - k->lit_rows.begin());
            const int n_items = seams ? i1 - i0 : 0, item_blocks = (n_items + 255) / 256, lit_blocks = (l1 - l0) * ((W + 255) / 256);
            if (item_blocks + lit_blocks > 0)
                hipLaunchKernelGGL(k->all_two ? k_temperature_edges<true> : k_temperature_edges<false>, dim3(item_blocks + lit_blocks), dim3(256), 0, ctx->stream,
                                   c, cur_v, nxt_v, land4_v, asr_i, k->dcos.f(), tc, (const int *)k->dshift.p, (const int *)k->dsrow.p,
                                   (const temp_rowrec *)k->drec.p, (const temp_item *)k->ditems.p + i0, n_items, item_blocks,
                                   l1 > l0 ? (const int *)k->dlitrows.p + l0 : nullptr, lo, hi, safe_lo, safe_hi);
        } else {
            hipLaunchKernelGGL(k_temperature, dim3((W + 255) / 256, nrows), dim3(256), 0, ctx->stream, c, cur_v, nxt_v, terr_v, asr_i,
                               k->dcos.f(), (const float4 *)k->dden.p, (const int *)k->ddentwo.p, tc, (const int *)k->dshift.p,
                               (const int *)k->dsrow.p, (const int *)k->dlit.p, (const int *)nullptr);
        }
        DMG_HIP(hipGetLastError());
        std::swap(k->cur, k->nxt);                   // terrain.swap(scratch2) (:148)
    }
    return DMG_OK;
}

int dmg_climate_get(dmg_climate *k, float *T)
{
    DMG_CHECK_ARG(k != nullptr && T != nullptr);
    DMG_HIP(hipSetDevice(k->ctx->device));
    // owned rows into the FULL-grid array (the whole field when there is one band)
    DMG_HIP(hipMemcpyAsync(T + (size_t)k->y0 * k->grid.W, k->cur + (size_t)k->gs * k->grid.W, (size_t)k->grid.W * (k->y1 - k->y0) * 4,
                           hipMemcpyDeviceToHost, k->ctx->stream));
    DMG_HIP(hipStreamSynchronize(k->ctx->stream));
    return DMG_OK;
}

// new heights under the same climate field (only their sign matters, :131-132): FULL-grid array, the band is sliced out
int dmg_climate_set_terrain(dmg_climate *k, const float *terrain)
{
    DMG_CHECK_ARG(k != nullptr && terrain != nullptr);
    DMG_HIP(hipSetDevice(k->ctx->device));
    DMG_HIP(hipMemcpyAsync(k->terr.p, terrain + k->voff(), (size_t)k->grid.W * k->L * 4, hipMemcpyHostToDevice, k->ctx->stream));
    if (k->land4.p) {
        const size_t n4 = (size_t)k->grid.W * k->L / 4;
        hipLaunchKernelGGL(k_land_mask4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, k->ctx->stream, k->terr.f(), (unsigned char *)k->land4.p, n4);
        DMG_HIP(hipGetLastError());
    }
    DMG_HIP(hipStreamSynchronize(k->ctx->stream));
    return DMG_OK;
}

// the same from a FULL-grid array on the climate's device, ordered on its stream (no host round trip, no synchronisation)
extern "C++" int climate_set_terrain_device(dmg_climate *k, const float *d_terrain)
{
    DMG_CHECK_ARG(k != nullptr && d_terrain != nullptr);
    DMG_HIP(hipSetDevice(k->ctx->device));
    DMG_HIP(hipMemcpyAsync(k->terr.p, d_terrain + k->voff(), (size_t)k->grid.W * k->L * 4, hipMemcpyDeviceToDevice, k->ctx->stream));
    if (k->land4.p) {
        const size_t n4 = (size_t)k->grid.W * k->L / 4;
        hipLaunchKernelGGL(k_land_mask4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, k->ctx->stream, k->terr.f(), (unsigned char *)k->land4.p, n4);
        DMG_HIP(hipGetLastError());
    }
    return DMG_OK;
}

// the same for a band whose terrain lives as a FULL-grid array on ANOTHER GPU (dmg_multi_world: tectonics and erosion run on the
// first band's device): a peer copy of the band's rows on the band's stream, which first waits for `ready` (the producer's event)
extern "C++" int climate_set_terrain_peer(dmg_climate *k, const float *d_full, int src_device, hipEvent_t ready)
{
    DMG_CHECK_ARG(k != nullptr && d_full != nullptr);
    DMG_HIP(hipSetDevice(k->ctx->device));
    if (ready) DMG_HIP(hipStreamWaitEvent(k->ctx->stream, ready, 0));
    const size_t bytes = (size_t)k->grid.W * k->L * 4;
    if (src_device == k->ctx->device) DMG_HIP(hipMemcpyAsync(k->terr.p, d_full + k->voff(), bytes, hipMemcpyDeviceToDevice, k->ctx->stream));
    else DMG_HIP(hipMemcpyPeerAsync(k->terr.p, k->ctx->device, d_full + k->voff(), src_device, bytes, k->ctx->stream));
    if (k->land4.p) {
        const size_t n4 = (size_t)k->grid.W * k->L / 4;
        hipLaunchKernelGGL(k_land_mask4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, k->ctx->stream, k->terr.f(), (unsigned char *)k->land4.p, n4);
        DMG_HIP(hipGetLastError());
    }
    return DMG_OK;
}

// ghost slabs of the temperature field, as dmg_ocean_halo
int dmg_climate_halo(dmg_climate *k, int side, int rows, void **send, void **recv, size_t *bytes)
{
    DMG_CHECK_ARG(k != nullptr && send != nullptr && recv != nullptr && bytes != nullptr && (side == DMG_SIDE_SOUTH || side == DMG_SIDE_NORTH));
    DMG_CHECK_ARG(rows >= 1 && rows <= std::max(k->ghost, 1));
    const size_t row = (size_t)k->grid.W;
    const int own = k->y1 - k->y0;
    *send = *recv = nullptr; *bytes = 0;
    if (side == DMG_SIDE_SOUTH && k->gs) { *send = k->cur + row * k->gs; *recv = k->cur + row * (k->gs - rows); *bytes = row * rows * 4; }
    else if (side == DMG_SIDE_NORTH && k->gn) { *send = k->cur + row * (k->gs + own - rows); *recv = k->cur + row * (k->gs + own); *bytes = row * rows * 4; }
    return DMG_OK;
}

int dmg_temperature_steps(dmg_ctx *ctx, const dmg_grid *grid, float *T, const float *terrain, const dmg_climate_params *params,
                          int first_step, int n_steps)
{
    DMG_CHECK_ARG(ctx != nullptr && T != nullptr && terrain != nullptr && n_steps >= 0);
    dmg_climate *k = nullptr;
    int rc = dmg_climate_create(ctx, grid, T, terrain, &k);
    if (rc) return rc;
    if (params) rc = dmg_climate_set_params(k, params);
    if (!rc) rc = dmg_climate_steps(k, first_step, n_steps);
    if (!rc) rc = dmg_climate_get(k, T);
    dmg_climate_destroy(k);
    return rc;
}

int dmg_erosion_pass(dmg_ctx *ctx, const dmg_grid *grid, float *h, const float *flow, const float *updrift,
                     float factor, float slope_exponent)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr && flow != nullptr && updrift != nullptr);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const int H = grid->H;
    const size_t n = (size_t)grid->W * H;
    std::vector<float> dxr(H), dyr(H);
    for (int y = 0; y < H; y++) dmg_row_pixelsize(*grid, y, &dxr[y], &dyr[y]);
    dev_buf dh, df, du, dout, ddx, ddy;
    DMG_ALLOC(dh, n * 4); DMG_ALLOC(df, n * 4); DMG_ALLOC(du, n * 4); DMG_ALLOC(dout, n * 4);
    if ((rc = upload_rows(ctx, ddx, dxr)) || (rc = upload_rows(ctx, ddy, dyr))) return rc;
    DMG_HIP(hipMemcpyAsync(dh.p, h, n * 4, hipMemcpyHostToDevice, ctx->stream));
    DMG_HIP(hipMemcpyAsync(df.p, flow, n * 4, hipMemcpyHostToDevice, ctx->stream));
    DMG_HIP(hipMemcpyAsync(du.p, updrift, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = erosion_device(ctx, *grid, topo, dh.f(), df.f(), du.f(), dout.f(), ddx.f(), ddy.f(), factor, slope_exponent))) return rc;
    DMG_HIP(hipMemcpyAsync(h, dout.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

__global__ void k_erosion_prepare(const float *__restrict__ img, float *__restrict__ updrift, float *__restrict__ height, float N, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = img[i];
    updrift[i] = __fdiv_rn(fmaxf(v, 0.0f), N);           // max(texture2D(img,st).r,0)/N  (cpufilter.cpp:52)
    height[i] = (v <= 0.0f) ? v : __fdiv_rn(v, N);       // :70-72
}

// cpufilter::run (:41-222), device resident: no readbacks between the flow solve and the erosion pass
struct dmg_erosion {
    dmg_ctx *ctx = nullptr;
    dmg_grid grid{};
    dmg_topology topo{};
    dev_buf terrain, height, next, updrift, flow, ddx, ddy;
    float *cur = nullptr, *nxt = nullptr;
    float exponent = 0.5f, slope_exponent = 1.0f, factor = 1.0f;
    int lakes = 0;
    long long rounds = 0;
};

int dmg_erosion_create(dmg_ctx *ctx, const dmg_grid *grid, const float *h, int N, float exponent, float slope_exponent, float factor,
                       int lakes, dmg_erosion **out)
{
    return erosion_create_from(ctx, grid, h, false, N, exponent, slope_exponent, factor, lakes, out);
}

// h: the terrain on the host, or (on_device) a W x H array on ctx's device, ordered on ctx's stream: a coupled driver
// (world.hip) hands the tectonics render target over without a host round trip
extern "C++" int erosion_create_from(dmg_ctx *ctx, const dmg_grid *grid, const float *h, bool on_device, int N, float exponent, float slope_exponent,
                                     float factor, int lakes, dmg_erosion **out)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr && N >= 1 && out != nullptr);
    *out = nullptr;
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const int H = grid->H;
    const size_t n = (size_t)grid->W * H;
    std::vector<float> dxr(H), dyr(H);
    for (int y = 0; y < H; y++) dmg_row_pixelsize(*grid, y, &dxr[y], &dyr[y]);
    std::unique_ptr<dmg_erosion> e(new (std::nothrow) dmg_erosion);
    if (!e) return dmg_fail(DMG_ENOMEM, "out of host memory");
    e->ctx = ctx; e->grid = *grid; e->topo = topo;
    e->exponent = exponent; e->slope_exponent = slope_exponent; e->factor = factor; e->lakes = lakes;
    DMG_ALLOC(e->terrain, n * 4); DMG_ALLOC(e->height, n * 4); DMG_ALLOC(e->next, n * 4); DMG_ALLOC(e->updrift, n * 4); DMG_ALLOC(e->flow, n * 4);
    if ((rc = upload_rows(ctx, e->ddx, dxr)) || (rc = upload_rows(ctx, e->ddy, dyr))) return rc;
    DMG_HIP(hipMemcpyAsync(e->terrain.p, h, n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_erosion_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, e->terrain.f(), e->updrift.f(),
                       e->height.f(), (float)N, n);
    DMG_HIP(hipGetLastError());
    if (!on_device) DMG_HIP(hipStreamSynchronize(ctx->stream));      // the caller's buffer may go away
    e->cur = e->height.f(); e->nxt = e->next.f();
    dmg_ctx_retain(ctx);
    *out = e.release();
    return DMG_OK;
}

void dmg_erosion_destroy(dmg_erosion *e)
{
    if (!e) return;
    dmg_ctx *ctx = e->ctx;
    delete e;
    dmg_ctx_release(ctx);
}

int dmg_erosion_rounds(dmg_erosion *e, int n_rounds)
{
    DMG_CHECK_ARG(e != nullptr && n_rounds >= 0);
    DMG_HIP(hipSetDevice(e->ctx->device));
    for (int i = 0; i < n_rounds; i++, e->rounds++) {      // :93
        // the first FlowFilter sees the unscaled terrain texture, every later one the eroded scaled heights (:95-108)
        const float *flow_in = (e->rounds == 0) ? e->terrain.f() : e->cur;
        int rc = flow_device(e->ctx, e->grid, e->topo, flow_in, nullptr, 0.5f, e->exponent, e->flow.f(), e->lakes);
        if (rc) return rc;
        rc = erosion_device(e->ctx, e->grid, e->topo, e->cur, e->flow.f(), e->updrift.f(), e->nxt, e->ddx.f(), e->ddy.f(), e->factor, e->slope_exponent);
        if (rc) return rc;
        std::swap(e->cur, e->nxt);
    }
    return DMG_OK;
}

extern "C++" const float *erosion_heights_device(const dmg_erosion *e) { return e->cur; }    // the scaled heights of the current round, on the device

int dmg_erosion_get(dmg_erosion *e, float *h)
{
    DMG_CHECK_ARG(e != nullptr && h != nullptr);
    DMG_HIP(hipSetDevice(e->ctx->device));
    DMG_HIP(hipMemcpyAsync(h, e->cur, (size_t)e->grid.W * e->grid.H * 4, hipMemcpyDeviceToHost, e->ctx->stream));
    DMG_HIP(hipStreamSynchronize(e->ctx->stream));
    return DMG_OK;
}

int dmg_erosion_run(dmg_ctx *ctx, const dmg_grid *grid, float *h, int N, int iters, float exponent,
                    float slope_exponent, float factor)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr && N >= 1 && iters >= 0);
    dmg_erosion *e = nullptr;
    int rc = dmg_erosion_create(ctx, grid, h, N, exponent, slope_exponent, factor, 0, &e);
    if (rc) return rc;
    rc = dmg_erosion_rounds(e, iters);
    if (!rc) rc = dmg_erosion_get(e, h);
    dmg_erosion_destroy(e);
    return rc;
}

// ThermalErosion::step (reference src/filter/ThermalErosion.cpp:25-142): 10 rounds of loss + gain, in place
int dmg_thermal_erosion(dmg_ctx *ctx, const dmg_grid *grid, float *h)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    canvas c = make_canvas(*grid, topo);
    const int H = grid->H;
    const size_t n = (size_t)grid->W * H;
    std::vector<float> dxr(H), dyr(H);
    for (int y = 0; y < H; y++) dmg_row_pixelsize(*grid, y, &dxr[y], &dyr[y]);
    dev_buf a, b, loss, ddx, ddy;
    DMG_ALLOC(a, n * 4); DMG_ALLOC(b, n * 4); DMG_ALLOC(loss, n * 4);
    if ((rc = upload_rows(ctx, ddx, dxr)) || (rc = upload_rows(ctx, ddy, dyr))) return rc;
    DMG_HIP(hipMemcpyAsync(a.p, h, n * 4, hipMemcpyHostToDevice, ctx->stream));
    float *cur = a.f(), *nxt = b.f();
    dim3 g((grid->W + 255) / 256, H);
    for (int i = 0; i < 10; i++) {                    // :26
        hipLaunchKernelGGL(k_thermal_loss, g, dim3(256), 0, ctx->stream, c, cur, ddx.f(), ddy.f(), loss.f());
        hipLaunchKernelGGL(k_thermal_gain, g, dim3(256), 0, ctx->stream, c, cur, loss.f(), ddx.f(), ddy.f(), nxt);
        DMG_HIP(hipGetLastError());
        std::swap(cur, nxt);                          // scratch2.swap(terrain) (:134)
    }
    DMG_HIP(hipMemcpyAsync(h, cur, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

// Morphological radius schedule (reference src/filter/Morphological.cpp:31-43)
static std::vector<float> morph_rlist(float radius)
{
    std::vector<float> r;
    int x = 1;
    while (radius >= 0) {
        if (x < radius) { radius -= x; r.push_back((float)x); x *= 2; }
        else { r.push_back(radius); break; }
    }
    std::sort(r.begin(), r.end());
    return r;
}

static int morph_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, float *d_field, float *d_tmp, float radius,
                        int use_max, float **result)
{
    canvas c = make_canvas(grid, topo);
    std::vector<float> fac(grid.H);
    for (int y = 0; y < grid.H; y++) fac[y] = 1 / glsl::cos(fabsf(dmg_row_lat(grid, y)));     // :53-54
    ring64 ring;
    for (int i = 0; i < 64; i++) { float a = 2 * 3.14159f * (float)i / 64.0f; ring.cs[i] = glsl::cos(a); ring.sn[i] = glsl::sin(a); }   // a 64-iteration shader loop: runtime sin/cos
    dev_buf dfac;
    int rc = upload_rows(ctx, dfac, fac);
    if (rc) return rc;
    float *cur = d_field, *nxt = d_tmp;
    dim3 g((grid.W + 255) / 256, grid.H);
    for (float r : morph_rlist(radius)) {             // Morphological::step (:75-85)
        hipLaunchKernelGGL(k_morph, g, dim3(256), 0, ctx->stream, c, cur, nxt, dfac.f(), r, use_max, ring);
        DMG_HIP(hipGetLastError());
        std::swap(cur, nxt);                          // scratch1.swap(target)
    }
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    *result = cur;
    return DMG_OK;
}

// Erode (op 0, "min") / Dilate (op 1, "max"): MorphologicalMenu::makeFilter (:24-27) + Morphological::step
int dmg_morphological(dmg_ctx *ctx, const dmg_grid *grid, float *field, float radius, int op)
{
    DMG_CHECK_ARG(ctx != nullptr && field != nullptr && radius >= 0.0f && (op == 0 || op == 1));
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)grid->W * grid->H;
    dev_buf a, b;
    DMG_ALLOC(a, n * 4); DMG_ALLOC(b, n * 4);
    DMG_HIP(hipMemcpyAsync(a.p, field, n * 4, hipMemcpyHostToDevice, ctx->stream));
    float *res = nullptr;
    if ((rc = morph_device(ctx, *grid, topo, a.f(), b.f(), radius, op, &res))) return rc;
    DMG_HIP(hipMemcpyAsync(field, res, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

__global__ void k_sub(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __fsub_rn(a[i], b[i]);
}

// MorphologicalGradient (:87-131): dilate((radius+1)/2) - erode(radius/2) of the same input
int dmg_morphological_gradient(dmg_ctx *ctx, const dmg_grid *grid, float *field, float radius)
{
    DMG_CHECK_ARG(ctx != nullptr && field != nullptr && radius >= 0.0f);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)grid->W * grid->H;
    dev_buf e0, e1, d0, d1;
    DMG_ALLOC(e0, n * 4); DMG_ALLOC(e1, n * 4); DMG_ALLOC(d0, n * 4); DMG_ALLOC(d1, n * 4);
    DMG_HIP(hipMemcpyAsync(e0.p, field, n * 4, hipMemcpyHostToDevice, ctx->stream));
    DMG_HIP(hipMemcpyAsync(d0.p, e0.p, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    float *er = nullptr, *di = nullptr;
    if ((rc = morph_device(ctx, *grid, topo, e0.f(), e1.f(), radius / 2, 0, &er))) return rc;          // :89
    if ((rc = morph_device(ctx, *grid, topo, d0.f(), d1.f(), (radius + 1) / 2, 1, &di))) return rc;    // :88
    float *out = (er == e0.f()) ? e1.f() : e0.f();
    hipLaunchKernelGGL(k_sub, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, di, er, out, n);   // :115-117
    DMG_HIP(hipGetLastError());
    DMG_HIP(hipMemcpyAsync(field, out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

}  // extern "C"

// =================================================================================================================
// Latitude-band forms of the generic filters (blur, temperature, erosion pass, thermal erosion
// shard by latitude band with 1-row / ceil(5.18 r)+1-row halos).  A dmg_field is one scalar field of one band, resident
// on the device with `ghost` rows towards each neighbouring band -- the same layout and the same dmg_*_halo slabs as a
// dmg_ocean band.  The kernels are the whole-grid kernels: they index a field by GRID row, so a band hands them
// base - (y0 - gs) * W ("virtual" base) and a row range (canvas::row0 + gridDim.y); the canvas keeps the full grid's
// topology, so wrap, pole reflection (lon += pi) and CLAMP_TO_EDGE resolve exactly as on one GPU.  A pass computes the
// band's OWNED rows from a field whose ghost rows the host has refreshed as deep as the pass reads
// (dmg_field_halo -> RCCL send/recv, or dmg_fields_exchange for bands living in one process).
// =================================================================================================================
struct dmg_field {
    dmg_ctx *ctx = nullptr;
    dmg_grid grid{};
    dmg_topology topo{};
    int rank = 0, count = 1, ghost = 0, y0 = 0, y1 = 0, gs = 0, gn = 0, L = 0;
    float *d[2] = { nullptr, nullptr };
    int cur = 0;
    dev_buf dpf, ddx, ddy;                           // per-row tables of the whole grid (tiny): blur's cos|lat|, pixelsize x / y
    dev_buf drec;                                    // the (0, r) blur pass' records of the band's rows (k_blur13_yrec)
    bool tables = false;
    float *virt(int k) const { return d[k] - (size_t)(y0 - gs) * grid.W; }
};

static int field_tables(dmg_field *f)
{
    if (f->tables) return DMG_OK;
    const int H = f->grid.H;
    std::vector<float> pf(H), dxr(H), dyr(H);
    for (int y = 0; y < H; y++) {
        pf[y] = glsl::cos(fabsf(dmg_row_lat(f->grid, y)));                       // BlurMenu.cpp:49
        dmg_row_pixelsize(f->grid, y, &dxr[y], &dyr[y]);
    }
    dmg_ctx *ctx = f->ctx;
    int rc;
    if ((rc = upload_rows(ctx, f->dpf, pf)) || (rc = upload_rows(ctx, f->ddx, dxr)) || (rc = upload_rows(ctx, f->ddy, dyr))) return rc;
    if (f->drec.alloc(ctx, BLUR_YREC_BYTES(f->y1 - f->y0)) != hipSuccess) return dmg_fail(DMG_ENOMEM, "field: no device memory for the blur's row records");
    f->tables = true;
    return DMG_OK;
}

static bool same_band(const dmg_field *a, const dmg_field *b)
{
    return a->grid.W == b->grid.W && a->grid.H == b->grid.H && a->y0 == b->y0 && a->y1 == b->y1 && a->gs == b->gs && a->gn == b->gn &&
           a->ctx->device == b->ctx->device;
}

extern "C" {

int dmg_field_create(dmg_ctx *ctx, const dmg_grid *grid, int band_rank, int band_count, int ghost, dmg_field **out)
{
    DMG_CHECK_ARG(ctx != nullptr && out != nullptr);
    *out = nullptr;
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_CHECK_ARG(band_count >= 1 && band_rank >= 0 && band_rank < band_count && ghost >= 0 && (band_count == 1 || ghost >= 1));
    DMG_CHECK_ARG(band_count == 1 || grid->H / band_count >= ghost);
    DMG_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<dmg_field> f(new (std::nothrow) dmg_field);
    if (!f) return dmg_fail(DMG_ENOMEM, "out of host memory");
    f->ctx = ctx; f->grid = *grid; f->topo = topo;
    f->rank = band_rank; f->count = band_count; f->ghost = ghost;
    f->y0 = dmg_band_row0(grid->H, band_count, band_rank);
    f->y1 = dmg_band_row0(grid->H, band_count, band_rank + 1);
    f->gs = band_rank > 0 ? ghost : 0;
    f->gn = band_rank < band_count - 1 ? ghost : 0;
    f->L = f->gs + (f->y1 - f->y0) + f->gn;
    const size_t bytes = (size_t)f->L * grid->W * 4;
    if (hipMalloc((void **)&f->d[0], bytes) != hipSuccess || hipMalloc((void **)&f->d[1], bytes) != hipSuccess) {
        if (f->d[0]) (void)hipFree(f->d[0]);
        return dmg_fail(DMG_ENOMEM, "hipMalloc of a %d-row band failed", f->L);
    }
    DMG_HIP(hipMemsetAsync(f->d[0], 0, bytes, ctx->stream));
    DMG_HIP(hipMemsetAsync(f->d[1], 0, bytes, ctx->stream));
    dmg_ctx_retain(ctx);
    *out = f.release();
    return DMG_OK;
}

void dmg_field_destroy(dmg_field *f)
{
    if (!f) return;
    dmg_ctx *ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (f->d[0]) (void)hipFree(f->d[0]);
    if (f->d[1]) (void)hipFree(f->d[1]);
    delete f;
    dmg_ctx_release(ctx);
}

// band + ghost rows out of a FULL-grid host array (Texture::uploadData side)
int dmg_field_set(dmg_field *f, const float *full)
{
    DMG_CHECK_ARG(f != nullptr && full != nullptr);
    DMG_HIP(hipSetDevice(f->ctx->device));
    DMG_HIP(hipMemcpyAsync(f->d[f->cur], full + (size_t)(f->y0 - f->gs) * f->grid.W, (size_t)f->L * f->grid.W * 4, hipMemcpyHostToDevice,
                           f->ctx->stream));
    DMG_HIP(hipStreamSynchronize(f->ctx->stream));
    return DMG_OK;
}

// owned rows into a FULL-grid host array (Texture::downloadDataRAW side)
int dmg_field_get(dmg_field *f, float *full)
{
    DMG_CHECK_ARG(f != nullptr && full != nullptr);
    DMG_HIP(hipSetDevice(f->ctx->device));
    DMG_HIP(hipMemcpyAsync(full + (size_t)f->y0 * f->grid.W, f->d[f->cur] + (size_t)f->gs * f->grid.W,
                           (size_t)(f->y1 - f->y0) * f->grid.W * 4, hipMemcpyDeviceToHost, f->ctx->stream));
    DMG_HIP(hipStreamSynchronize(f->ctx->stream));
    return DMG_OK;
}

int dmg_field_halo(dmg_field *f, int side, int rows, void **send, void **recv, size_t *bytes)
{
    DMG_CHECK_ARG(f != nullptr && send != nullptr && recv != nullptr && bytes != nullptr && (side == DMG_SIDE_SOUTH || side == DMG_SIDE_NORTH));
    DMG_CHECK_ARG(rows >= 1 && rows <= std::max(f->ghost, 1));
    const size_t row = (size_t)f->grid.W;
    const int own = f->y1 - f->y0;
    float *base = f->d[f->cur];
    *send = *recv = nullptr; *bytes = 0;
    if (side == DMG_SIDE_SOUTH && f->gs) { *send = base + row * f->gs; *recv = base + row * (f->gs - rows); *bytes = row * rows * 4; }
    else if (side == DMG_SIDE_NORTH && f->gn) { *send = base + row * (f->gs + own - rows); *recv = base + row * (f->gs + own); *bytes = row * rows * 4; }
    return DMG_OK;
}

// One halo refresh among the n bands of a field that live in THIS process (bands[i] = band i of n, any devices):
// every band pulls its ghost rows from its neighbours' owned rows with peer copies on its own stream, ordered by events.
int dmg_fields_exchange(dmg_field *const *bands, int n, int rows)
{
    DMG_CHECK_ARG(bands != nullptr && n >= 1 && rows >= 1);
    if (n == 1) return DMG_OK;
    for (int i = 0; i < n; i++) DMG_CHECK_ARG(bands[i] != nullptr && bands[i]->rank == i && bands[i]->count == n);   // before any event exists
    std::vector<hipEvent_t> done(n, nullptr), copied(n, nullptr);
    int rc = DMG_OK;
    auto cleanup = [&] { for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e); for (hipEvent_t e : copied) if (e) (void)hipEventDestroy(e); };
    for (int i = 0; i < n && !rc; i++) {
        if (hipSetDevice(bands[i]->ctx->device) != hipSuccess || hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess || hipEventRecord(done[i], bands[i]->ctx->stream) != hipSuccess)
            rc = dmg_fail(DMG_EHIP, "dmg_fields_exchange: event set-up failed");
    }
    for (int i = 0; i < n && !rc; i++) {
        dmg_field *f = bands[i];
        (void)hipSetDevice(f->ctx->device);
        for (int side = 0; side < 2 && !rc; side++) {
            const int j = side == DMG_SIDE_SOUTH ? i - 1 : i + 1;
            if (j < 0 || j >= n) continue;
            void *send, *recv, *nsend, *nrecv; size_t bytes, nbytes;
            if ((rc = dmg_field_halo(f, side, rows, &send, &recv, &bytes)) || (rc = dmg_field_halo(bands[j], side ^ 1, rows, &nsend, &nrecv, &nbytes))) break;
            if (hipStreamWaitEvent(f->ctx->stream, done[j], 0) != hipSuccess ||
                hipMemcpyPeerAsync(recv, f->ctx->device, nsend, bands[j]->ctx->device, bytes, f->ctx->stream) != hipSuccess)
                rc = dmg_fail(DMG_EHIP, "dmg_fields_exchange: peer copy failed");
        }
        if (!rc && hipEventRecord(copied[i], f->ctx->stream) != hipSuccess) rc = dmg_fail(DMG_EHIP, "hipEventRecord failed");
    }
    for (int i = 0; i < n && !rc; i++)
        for (int j : { i - 1, i + 1 })
            if (j >= 0 && j < n && hipStreamWaitEvent(bands[i]->ctx->stream, copied[j], 0) != hipSuccess) rc = dmg_fail(DMG_EHIP, "hipStreamWaitEvent failed");
    for (int i = 0; i < n; i++) { (void)hipSetDevice(bands[i]->ctx->device); (void)hipStreamSynchronize(bands[i]->ctx->stream); }   // the events go away
    cleanup();
    return rc;
}

// Blur::Blur's radius list (BlurMenu.cpp:27,79-94): Blur::step then runs, per radius, a vertical and a horizontal pass
int dmg_blur_schedule(float radius, float *r_out, int cap)
{
    if (!(radius > 0.0f) || r_out == nullptr || cap <= 0) return dmg_fail(DMG_EINVAL, "dmg_blur_schedule: bad argument");
    std::vector<float> rl = blur_rlist(radius);
    if ((int)rl.size() > cap) return dmg_fail(DMG_EINVAL, "dmg_blur_schedule: %zu radii, room for %d", rl.size(), cap);
    for (size_t i = 0; i < rl.size(); i++) r_out[i] = rl[i];
    return (int)rl.size();
}

// ghost rows a pass of radius r reads: the outermost of the 13 taps sits 5.18 r rows away (blur13, BlurMenu.cpp:54-60), the
// bilinear footprint adds one; a horizontal pass only its own row's bilinear neighbour
int dmg_field_blur_halo(float r, int vertical) { return vertical ? (int)ceilf(5.1765f * r) + 1 : 1; }

// one pass of Blur::step on the band: (0, r) when vertical, (r, 0) otherwise; the result becomes the current buffer
int dmg_field_blur_pass(dmg_field *f, float r, int vertical)
{
    DMG_CHECK_ARG(f != nullptr && r > 0.0f);
    if (f->count > 1 && dmg_field_blur_halo(r, vertical) > f->ghost)
        return dmg_fail(DMG_EINVAL, "blur pass of radius %g reads %d rows beyond the band, %d ghost rows kept", r, dmg_field_blur_halo(r, vertical), f->ghost);
    DMG_HIP(hipSetDevice(f->ctx->device));
    int rc = field_tables(f);
    if (rc) return rc;
    canvas c = make_canvas(f->grid, f->topo);
    c.row0 = f->y0;
    dim3 g((f->grid.W + 255) / 256, f->y1 - f->y0);
    launch_blur13(f->ctx->stream, g, c, (const float *)f->virt(f->cur), f->virt(f->cur ^ 1), f->dpf.f(), vertical ? 0.0f : r, vertical ? r : 0.0f, f->drec.p);
    DMG_HIP(hipGetLastError());
    f->cur ^= 1;
    return DMG_OK;
}

// erosion / uplift pass of cpufilter::run (cpufilter.cpp:106-214) on the band: h needs 1 current ghost row; flow and
// updrift are read at the cell only
int dmg_field_erosion_pass(dmg_field *h, const dmg_field *flow, const dmg_field *updrift, float factor, float slope_exponent)
{
    DMG_CHECK_ARG(h != nullptr && flow != nullptr && updrift != nullptr && same_band(h, flow) && same_band(h, updrift));
    DMG_HIP(hipSetDevice(h->ctx->device));
    int rc = field_tables(h);
    if (rc) return rc;
    canvas c = make_canvas(h->grid, h->topo);
    c.row0 = h->y0;
    const float SLOPE = tanf(DMG_PI / 2.0f / 3.0f), pow01 = glsl::pow(0.1f, slope_exponent);     // as erosion_device
    dim3 g((h->grid.W + 255) / 256, h->y1 - h->y0);
    hipLaunchKernelGGL(k_erosion, g, dim3(256), 0, h->ctx->stream, c, (const float *)h->virt(h->cur), (const float *)flow->virt(flow->cur),
                       (const float *)updrift->virt(updrift->cur), h->virt(h->cur ^ 1), h->ddx.f(), h->ddy.f(), factor, slope_exponent, SLOPE, pow01);
    DMG_HIP(hipGetLastError());
    h->cur ^= 1;
    return DMG_OK;
}

// one of ThermalErosion::step's ten rounds (ThermalErosion.cpp:25-142) on the band: talus loss on the owned rows and one
// row beyond them, gain on the owned rows -- h needs 2 current ghost rows, `loss` is scratch of the same band
int dmg_field_thermal_round(dmg_field *h, dmg_field *loss)
{
    DMG_CHECK_ARG(h != nullptr && loss != nullptr && same_band(h, loss));
    if (h->count > 1 && h->ghost < 2) return dmg_fail(DMG_EINVAL, "a thermal-erosion round reads 2 rows beyond the band, %d ghost rows kept", h->ghost);
    DMG_HIP(hipSetDevice(h->ctx->device));
    int rc = field_tables(h);
    if (rc) return rc;
    canvas c = make_canvas(h->grid, h->topo);
    const int lo = h->gs ? h->y0 - 1 : h->y0, hi = h->gn ? h->y1 + 1 : h->y1;
    c.row0 = lo;
    hipLaunchKernelGGL(k_thermal_loss, dim3((h->grid.W + 255) / 256, hi - lo), dim3(256), 0, h->ctx->stream, c, (const float *)h->virt(h->cur),
                       h->ddx.f(), h->ddy.f(), loss->virt(loss->cur));
    c.row0 = h->y0;
    hipLaunchKernelGGL(k_thermal_gain, dim3((h->grid.W + 255) / 256, h->y1 - h->y0), dim3(256), 0, h->ctx->stream, c, (const float *)h->virt(h->cur),
                       (const float *)loss->virt(loss->cur), h->ddx.f(), h->ddy.f(), h->virt(h->cur ^ 1));
    DMG_HIP(hipGetLastError());
    h->cur ^= 1;                                      // scratch2.swap(terrain) (:134)
    return DMG_OK;
}

}  // extern "C"
