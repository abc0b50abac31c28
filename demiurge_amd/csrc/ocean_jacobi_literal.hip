// The Jacobi sweep evaluated per fragment exactly as the reference's GLSL is evaluated on Mesa llvmpipe, including the
// two things of that GL implementation that show on grids that are not powers of two:
//   * `st` interpolated by the rasteriser through two fp32 plane equations (tools/st_probe.py), and
//   * GL_LINEAR taps with float weights, so a tap an ulp off the texel centre blends the neighbour in.
// Opt-in (dmg_ocean_params.quirks & DMG_QUIRK_LLVMPIPE_RASTER): for users who need the pixels of the software GL path on
// e.g. the reference's default 1000x500 canvas.  On power-of-two grids it gives the bits of the fast kernels, which is
// what they are tested against; it does ~60 fp32 divisions and one cos per cell and is not a performance path.
#include "ocean_internal.h"
#include "canvas.h"

__device__ __forceinline__ void d_st_raster(const canvas &c, int x, int y, float *sx, float *sy)
{
    const float Wf = (float)c.W, Hf = (float)c.H;
    const float ooa = __fdiv_rn(1.0f, __fmul_rn(Wf, Hf));
    const float d = __fmul_rn(Hf, ooa), e = __fmul_rn(Wf, ooa);
    const bool lower = ((double)x + 0.5) / c.W + ((double)y + 0.5) / c.H < 1.0;
    const float a0x = lower ? __fsub_rn(0.0f, __fmul_rn(d, -0.5f)) : __fsub_rn(1.0f, __fmul_rn(d, __fsub_rn(Wf, 0.5f)));
    const float a0y = lower ? __fsub_rn(0.0f, __fmul_rn(e, -0.5f)) : __fsub_rn(1.0f, __fmul_rn(e, __fsub_rn(Hf, 0.5f)));
    *sx = __fmaf_rn(d, (float)x, a0x);
    *sy = __fmaf_rn(e, (float)y, a0y);
}
// pixelsize(st) (src/Shader.h:71-79)
__device__ __forceinline__ void d_pixelsize(const canvas &c, float sy, float *dx, float *dy)
{
    const float lat = __fadd_rn(__fmul_rn(sy, __fsub_rn(c.c1, c.c0)), c.c0);
    const float two_pi = 2.0f * DMG_PI;
    *dx = __fdiv_rn(__fdiv_rn(__fmul_rn(__fmul_rn(__fsub_rn(c.c3, c.c2), glsl::cos(lat)), c.circumference), two_pi), (float)c.W);
    *dy = __fdiv_rn(__fdiv_rn(__fmul_rn(__fsub_rn(c.c1, c.c0), c.circumference), two_pi), (float)c.H);
}

// OceanCurrents.cpp:450-481, statement by statement
__global__ __launch_bounds__(256) void k_jacobi_literal(canvas c, const float *__restrict__ img, const float *__restrict__ divw,
                                                        const float *__restrict__ p, float *__restrict__ out, float scale)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= c.W) return;
    const int W = c.W, H = c.H;
    float sx, sy, dx, dy;
    d_st_raster(c, x, y, &sx, &sy);
    d_pixelsize(c, sy, &dx, &dy);
    const float wx = __fdiv_rn(dx, scale), wy = __fdiv_rn(dy, scale);
    const float pw2x = __fmul_rn(wx, wx), pw2y = __fmul_rn(wy, wy);                       // pow(v, vec2(2,2))
    const float Beta = __fmul_rn(2.0f, __fadd_rn(__fdiv_rn(1.0f, pw2x), __fdiv_rn(1.0f, pw2y)));   // :479
    float nx, ny, sx2, sy2, ex, ey, wx2, wy2;
    d_offset(c, sx, sy, 0.0f, 1.0f, &nx, &ny);
    d_offset(c, sx, sy, 0.0f, -1.0f, &sx2, &sy2);
    d_offset(c, sx, sy, 1.0f, 0.0f, &ex, &ey);
    d_offset(c, sx, sy, -1.0f, 0.0f, &wx2, &wy2);
    float pN = d_fetch_linear(p, W, H, nx, ny), pS = d_fetch_linear(p, W, H, sx2, sy2);
    float pE = d_fetch_linear(p, W, H, ex, ey), pW = d_fetch_linear(p, W, H, wx2, wy2);
    const float pC = d_fetch_linear(p, W, H, sx, sy);
    if (d_fetch_nearest(img, W, H, nx, ny) > 0.0f) pN = pC;                                // :470-473
    if (d_fetch_nearest(img, W, H, sx2, sy2) > 0.0f) pS = pC;
    if (d_fetch_nearest(img, W, H, ex, ey) > 0.0f) pE = pC;
    if (d_fetch_nearest(img, W, H, wx2, wy2) > 0.0f) pW = pC;
    const float bC = d_fetch_linear(divw, W, H, sx, sy);
    float fc = __fdiv_rn(__fsub_rn(__fadd_rn(__fdiv_rn(__fadd_rn(pW, pE), pw2x), __fdiv_rn(__fadd_rn(pS, pN), pw2y)), bC), Beta);   // :480
    if (d_fetch_nearest(img, W, H, sx, sy) > 0.0f) fc = 0.0f;                              // :481
    out[(size_t)y * W + x] = fc;
}

// one sweep, whole grid (single band only), from the current pressure buffer into the other one
int ocean_launch_literal(dmg_ocean *o)
{
    if (o->count != 1) return dmg_fail(DMG_EINVAL, "DMG_QUIRK_LLVMPIPE_RASTER is a single-band path");
    canvas c = make_canvas(o->grid, o->topo);
    dim3 g((o->grid.W + 255) / 256, o->grid.H);
    hipLaunchKernelGGL(k_jacobi_literal, g, dim3(256), 0, o->ctx->stream, c, o->img, o->divw, o->p[o->cur], o->p[o->cur ^ 1], o->prm.scale);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}
