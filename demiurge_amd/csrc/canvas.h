// Device-side restatement of the GLSL canvas helpers shared by filters.hip and flow.hip: offset()
// (reference src/Shader.h:82-98) and the texture unit (GL_LINEAR / GL_NEAREST + CLAMP_TO_EDGE, src/Texture.cpp:29-33).
#pragma once
#include "dmg_internal.h"

struct canvas {
    int W, H;
    float c0, c1, c2, c3, circumference;
    int lonwrap, spole, npole;
    int row0;                  // first grid row a launch computes (blockIdx.y = 0): 0 for a whole grid, a band's row range otherwise
};

static inline canvas make_canvas(const dmg_grid &g, const dmg_topology &t)
{
    canvas c;
    c.W = g.W; c.H = g.H;
    c.c0 = g.coords[0]; c.c1 = g.coords[1]; c.c2 = g.coords[2]; c.c3 = g.coords[3];
    c.circumference = g.circumference;
    c.lonwrap = t.lonwrap; c.spole = t.spole; c.npole = t.npole;
    c.row0 = 0;
    return c;
}

__device__ __forceinline__ float d_mod(float x, float y) { return __fsub_rn(x, __fmul_rn(y, floorf(__fdiv_rn(x, y)))); }

// mod(lon + 2*M_PI, 2*M_PI) - M_PI (Shader.h:88,93) with its constants combined the way Mesa's algebraic pass combines
// them: (lon + pi) - 2pi*floor((lon + 2pi)/2pi).  Bit-equal to llvmpipe on pole-crossing taps with fractional offsets
// (blur); the literal form is not (93 %).
__device__ __forceinline__ float d_pole_lon(float lon)
{
    const float two_pi = 2.0f * DMG_PI;
    return __fsub_rn(__fadd_rn(lon, DMG_PI), __fmul_rn(two_pi, floorf(__fdiv_rn(__fadd_rn(lon, two_pi), two_pi))));
}

// offset(p, dp, resolution)  (reference src/Shader.h:82-98), statement by statement in fp32
__device__ __forceinline__ void d_offset(const canvas &c, float px, float py, float dpx, float dpy, float *ox, float *oy)
{
    px = __fadd_rn(px, __fdiv_rn(dpx, (float)c.W));
    py = __fadd_rn(py, __fdiv_rn(dpy, (float)c.H));
    if (c.lonwrap) px = d_mod(__fadd_rn(px, 1.0f), 1.0f);
    if (c.spole && py < 0.0f) {
        py = -py;
        px = d_pole_lon(__fadd_rn(__fmul_rn(px, __fsub_rn(c.c3, c.c2)), c.c2));
        px = __fdiv_rn(__fsub_rn(px, c.c2), __fsub_rn(c.c3, c.c2));
    }
    if (c.npole && py > 1.0f) {
        py = __fsub_rn(2.0f, py);
        px = d_pole_lon(__fadd_rn(__fmul_rn(px, __fsub_rn(c.c3, c.c2)), c.c2));
        px = __fdiv_rn(__fsub_rn(px, c.c2), __fsub_rn(c.c3, c.c2));
    }
    if (c.lonwrap) px = d_mod(__fadd_rn(px, 1.0f), 1.0f);
    *ox = px; *oy = py;
}

__device__ __forceinline__ float d_fetch_nearest(const float *t, int W, int H, float u, float v)
{
    int i = min(max((int)floorf(__fmul_rn(u, (float)W)), 0), W - 1);
    int j = min(max((int)floorf(__fmul_rn(v, (float)H)), 0), H - 1);
    return t[(size_t)j * W + i];
}

__device__ __forceinline__ float d_fetch_linear(const float *t, int W, int H, float u, float v)
{
    float fu = __fsub_rn(__fmul_rn(u, (float)W), 0.5f), fv = __fsub_rn(__fmul_rn(v, (float)H), 0.5f);
    float flu = floorf(fu), flv = floorf(fv);
    float wx = __fsub_rn(fu, flu), wy = __fsub_rn(fv, flv);
    int i0 = min(max((int)flu, 0), W - 1), i1 = min(max((int)flu + 1, 0), W - 1);
    int j0 = min(max((int)flv, 0), H - 1), j1 = min(max((int)flv + 1, 0), H - 1);
    float t00 = t[(size_t)j0 * W + i0], t10 = t[(size_t)j0 * W + i1], t01 = t[(size_t)j1 * W + i0], t11 = t[(size_t)j1 * W + i1];
    float a = fmaf(wx, __fsub_rn(t10, t00), t00);        // llvmpipe's lerp is one fused multiply-add (lp_build_lerp)
    float b = fmaf(wx, __fsub_rn(t11, t01), t01);
    return fmaf(wy, __fsub_rn(b, a), a);
}

__device__ __forceinline__ float d_stx(const canvas &c, int x) { return __fdiv_rn(__fadd_rn((float)x, 0.5f), (float)c.W); }
__device__ __forceinline__ float d_sty(const canvas &c, int y) { return __fdiv_rn(__fadd_rn((float)y, 0.5f), (float)c.H); }


// nearest texel reached by offset(st,(dx,dy)) for integer steps: wrap or clamp in x, pole reflection or clamp in y
__device__ __forceinline__ int d_step(const canvas &c, int x, int y, int dx, int dy)
{
    int xx = x + dx, yy = y + dy;
    if (c.lonwrap) { if (xx >= c.W) xx -= c.W; if (xx < 0) xx += c.W; }
    else xx = min(max(xx, 0), c.W - 1);
    if (yy < 0) { yy = 0; if (c.spole) { xx += c.W / 2; if (xx >= c.W) xx -= c.W; } }
    if (yy >= c.H) { yy = c.H - 1; if (c.npole) { xx += c.W / 2; if (xx >= c.W) xx -= c.W; } }
    return yy * c.W + xx;
}

// device-resident Gaussian blur (Blur::Blur + Blur::step, reference src/filter/BlurMenu.cpp:24-117): blurs d_field
// (W*H floats) in place, d_tmp is W*H floats of workspace.  Asynchronous on the context's stream.
int blur_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, float *d_field, float *d_tmp, float radius);

// device-resident flow accumulation (flow.hip) and erosion pass (filters.hip), used by dmg_erosion_run
int flow_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, const float *d_h, const float *d_sel,
                float preblur, float exponent, float *d_flow, int lakes = 0);
int erosion_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, const float *d_h, const float *d_flow,
                   const float *d_updrift, float *d_out, const float *d_dx, const float *d_dy, float factor, float slope_exponent);

namespace {
// a temporary device buffer of one call, drawn from (and returned to) the context's scratch pool
struct dev_buf {
    void *p = nullptr;
    dmg_ctx *owner = nullptr;
    size_t bytes = 0;
    ~dev_buf() { reset(); }
    void reset() { if (p) dmg_scratch_free(owner, p, bytes); p = nullptr; bytes = 0; }     // (before a second upload into the same buffer)
    // grid-sized fields (and everything else whose size does not change from call to call): the size asked for
    hipError_t alloc(dmg_ctx *ctx, size_t n)
    {
        owner = ctx;
        bytes = (n + 255) & ~(size_t)255;
        return dmg_scratch_alloc(ctx, bytes, &p);
    }
    // sizes that depend on the data (lakes, passes, lake pairs: different every erosion round) are rounded up to three
    // significant bits, so that a round finds the blocks the round before it parked: the pool matches sizes exactly, and a
    // miss is a hipMalloc now and a hipFree (which synchronises) later.  Not for the resident fields (up to 25 %
    // on top of a 16384 x 1040 band's 68 MB per buffer bought nothing)
    hipError_t alloc_var(dmg_ctx *ctx, size_t n)
    {
        owner = ctx;
        bytes = (n + 255) & ~(size_t)255;
        if (bytes > 1024) {
            int m = 63 - __builtin_clzll((unsigned long long)bytes);
            const size_t step = (size_t)1 << (m - 2);
            bytes = (bytes + step - 1) & ~(step - 1);
        }
        return dmg_scratch_alloc(ctx, bytes, &p);
    }
    float *f() const { return (float *)p; }
    int *i() const { return (int *)p; }
};
}  // namespace

#define DMG_ALLOC(buf, bytes)                                                                      \
    do {                                                                                           \
        hipError_t _e = (buf).alloc(ctx, bytes);                                                   \
        if (_e != hipSuccess) return dmg_fail(DMG_ENOMEM, "hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(_e)); \
    } while (0)

#define DMG_ALLOC_VAR(buf, bytes)                                                                  \
    do {                                                                                           \
        hipError_t _e = (buf).alloc_var(ctx, bytes);                                               \
        if (_e != hipSuccess) return dmg_fail(DMG_ENOMEM, "hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(_e)); \
    } while (0)

template <typename T>
static inline int upload_vec(dmg_ctx *ctx, dev_buf &d, const std::vector<T> &v)
{
    DMG_ALLOC_VAR(d, v.size() * sizeof(T) + 16);
    DMG_HIP(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));      // `v` is a host temporary
    return DMG_OK;
}
