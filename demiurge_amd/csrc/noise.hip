// Gradient-noise terrain generator (reference src/filter/GradientNoise.cpp) on gfx950: 3-D simplex noise with analytic
// gradient (the Ashima Arts / stegu webgl-noise `snoise` the reference embeds, :66-175), seven octave schemes
// (:181-437), filter::blendMode() (Filter.cpp:170-207), seed offsets from MT19937 (:489-491).
//
// One thread per texel, no memory traffic beyond 4-12 B/texel: pure VALU (8 octaves x ~250 fp32 ops).  The kernel is
// templated on the noise mode so each instantiation carries only its own octave loop.
//
// Arithmetic contract: the octave loop adds seed offsets of up to 1e4 to a position on the unit sphere, i.e. it
// quantises the position to ~1e-3; one ulp of difference before that sum moves a texel's noise by ~1e-3 of the range.
// So the expression ORDER is part of the algorithm.  It follows what Mesa's compiler makes of the GLSL (pinned by
// probe shaders, see oracle/noise_oracle.c): dot() accumulates from the last component, dot(v, vec3(c)) becomes
// ((v.z+v.y)+v.x)*c, `p + cross(p,d)` becomes (p + a*b) - c*d, a single-use `1 + dot(v,v)` takes the constant
// innermost, smoothstep is t*(t*(3-2t)), `x*x*5` is x*(x*5); sin/cos are the Cephes octant scheme (tectonics.hip).
#include <algorithm>
#include <cstdint>
#include "canvas.h"
#include "glsl_math.h"
#include "ocean_internal.h"

namespace {

struct v3 { float x, y, z; };
__device__ __forceinline__ v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator+(v3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
__device__ __forceinline__ float dot_splat(v3 a, float c) { return ((a.z + a.y) + a.x) * c; }
__device__ __forceinline__ float one_plus_dot(v3 a) { return ((1.0f + a.z * a.z) + a.y * a.y) + a.x * a.x; }
__device__ __forceinline__ float length(v3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ v3 add_cross(v3 p, v3 d)
{ return V3((p.x + p.y * d.z) - p.z * d.y, (p.y + p.z * d.x) - p.x * d.z, (p.z + p.x * d.y) - p.y * d.x); }
__device__ __forceinline__ float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
__device__ __forceinline__ float mod289(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
__device__ __forceinline__ float permute(float x) { return mod289(((x * 34.0f) + 1.0f) * x); }

__device__ __forceinline__ void n_sincos(float x, float *s, float *c) { glsl::sincos(x, s, c); }   // glsl_math.h

// snoise(v, out gradient) (:97-175)
__device__ float snoise(v3 v, v3 *gradient)
{
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    float s = dot_splat(v, Cy);
    v3 i = V3(floorf(v.x + s), floorf(v.y + s), floorf(v.z + s));
    v3 x0 = (v - i) + dot_splat(i, Cx);
    v3 g = V3(step(x0.y, x0.x), step(x0.z, x0.y), step(x0.x, x0.z));
    v3 l = V3(1.0f - g.x, 1.0f - g.y, 1.0f - g.z);
    v3 i1 = V3(fminf(g.x, l.z), fminf(g.y, l.x), fminf(g.z, l.y));
    v3 i2 = V3(fmaxf(g.x, l.z), fmaxf(g.y, l.x), fmaxf(g.z, l.y));
    v3 xs[4];
    xs[0] = x0;
    xs[1] = (x0 - i1) + Cx;
    xs[2] = (x0 - i2) + Cy;
    xs[3] = V3(x0.x - 0.5f, x0.y - 0.5f, x0.z - 0.5f);
    i = V3(mod289(i.x), mod289(i.y), mod289(i.z));
    const float oz[4] = { 0.0f, i1.z, i2.z, 1.0f }, oy[4] = { 0.0f, i1.y, i2.y, 1.0f }, ox[4] = { 0.0f, i1.x, i2.x, 1.0f };
    const float n_ = 0.142857142857f;
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
    v3 gx = V3(0.0f, 0.0f, 0.0f), gp = V3(0.0f, 0.0f, 0.0f);
    float m4[4], pdotx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float p = permute((permute((permute(i.z + oz[k]) + i.y) + oy[k]) + i.x) + ox[k]);
        float j = p - 49.0f * floorf(p * nsz * nsz);
        float x_ = floorf(j * nsz);
        float y_ = floorf(j - 7.0f * x_);
        float X = x_ * nsx + nsy, Y = y_ * nsx + nsy;
        float h = 1.0f - fabsf(X) - fabsf(Y);
        float sh = -step(h, 0.0f);
        v3 P = V3(X + (floorf(X) * 2.0f + 1.0f) * sh, Y + (floorf(Y) * 2.0f + 1.0f) * sh, h);
        P = P * (1.79284291400159f - 0.85373472095314f * dot(P, P));
        float m = fmaxf(0.6f - dot(xs[k], xs[k]), 0.0f);
        float m2 = m * m;
        m4[k] = m2 * m2;
        pdotx[k] = dot(P, xs[k]);
        float temp = m2 * m * pdotx[k];
        gx = k == 0 ? xs[k] * temp : gx + xs[k] * temp;
        gp = k == 0 ? P * m4[k] : gp + P * m4[k];
    }
    *gradient = (gx * -8.0f + gp) * 42.0f;
    return 42.0f * (((m4[3] * pdotx[3] + m4[2] * pdotx[2]) + m4[1] * pdotx[1]) + m4[0] * pdotx[0]);
}

// rotation_matrix(theta, u) * p (Shader.h:35-41)
__device__ v3 rotate(float theta, v3 u, v3 p)
{
    float s, c;
    n_sincos(theta, &s, &c);
    float oc = 1.0f - c;
    v3 c0 = V3(c + u.x * u.x * oc, u.y * u.x * oc + u.z * s, u.z * u.x * oc - u.y * s);
    v3 c1 = V3(u.x * u.y * oc - u.z * s, c + u.y * u.y * oc, u.z * u.y * oc + u.x * s);
    v3 c2 = V3(u.x * u.z * oc + u.y * s, u.y * u.z * oc - u.x * s, c + u.z * u.z * oc);
    return (c0 * p.x + c1 * p.y) + c2 * p.z;
}

// dot(tmp,p)/length(p) * p/length(p)
__device__ __forceinline__ v3 radial_of(v3 tmp, v3 p) { float len = length(p); return (p * (dot(tmp, p) / len)) / len; }
__device__ __forceinline__ v3 tangential(v3 tmp, v3 p) { return tmp - radial_of(tmp, p); }

// the domain-warp prologue of DEFAULT / SWISS / Jordan / Plateaus
__device__ v3 domain_warp(v3 p, v3 at, float warp)
{
    v3 tmp;
    snoise(at, &tmp);
    tmp = tangential(tmp, p);
    v3 u = p + tmp;
    u = u / dot(u, u);
    return rotate(warp * 0.1f * length(tmp), u, p);
}

__device__ __forceinline__ float smoothstep(float e0, float e1, float x)
{
    float t = fminf(fmaxf((x - e0) / (e1 - e0), 0.0f), 1.0f);
    return t * (t * (3.0f - 2.0f * t));
}

struct noise_args {
    float scale, lacunarity, persistence, lo, hi, warp;
    int octaves, blend;
    float so[3];
};

enum { N_DEFAULT, N_RIDGED, N_BILLOWY, N_IQ, N_SWISS, N_JORDAN, N_PLATEAUS };

template <int MODE>
__device__ float noise_value(const noise_args &q, float lon, float lat)
{
    const v3 so = V3(q.so[0], q.so[1], q.so[2]);
    float so_, co_, sa, ca;
    n_sincos(lon, &so_, &co_);
    v3 p;
    if (MODE == N_DEFAULT) {                                     // scale*spheric_to_cartesian(...) :183
        n_sincos(lat, &sa, &ca);
        p = V3(ca * co_, ca * so_, sa) * q.scale;
    } else {                                                     // vec3(sin(M_PI/2-geo.y)*cos(geo.x), ...) :227
        n_sincos(DMG_PI / 2.0f - lat, &sa, &ca);
        p = V3(sa * co_, sa * so_, ca);
        if (MODE == N_RIDGED || MODE == N_BILLOWY || MODE == N_IQ) p = p * q.scale;
    }
    float cur = 1.0f, total = 0.0f, fc = 0.0f;
    v3 tmp;
    if (MODE == N_DEFAULT) {                                     // :182-221
        p = domain_warp(p, p, q.warp);
        for (int i = 0; i < q.octaves; i++) {
            fc += snoise(p + so * (float)(i + 1), &tmp) * cur;
            p = p * q.lacunarity;
            total += cur;
            cur *= q.persistence;
        }
        fc /= total;
        return (fc + 1.0f) * 0.5f * (q.hi - q.lo) + q.lo;
    }
    if (MODE == N_RIDGED || MODE == N_BILLOWY) {                 // :224-263
        for (int i = 0; i < q.octaves; i++) {
            float n = fabsf(snoise(p + so, &tmp));
            fc += (MODE == N_RIDGED ? 1.0f - n : n) * cur;
            p = p * q.lacunarity;
            total += cur;
            cur *= q.persistence;
        }
        fc /= total;
        return fc * (q.hi - q.lo) + q.lo;
    }
    if (MODE == N_IQ) {                                          // :266-288
        v3 dsum = V3(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < q.octaves; i++) {
            float n = snoise(p + so, &tmp) * cur;
            dsum = dsum + (tmp - radial_of(tmp, p));
            float den = 1.0f + dot(dsum, dsum);
            fc += n / den;
            p = p * q.lacunarity;
            total += cur / den;
            cur *= q.persistence;
        }
        fc /= total;
        return (fc + 1.0f) * 0.5f * (q.hi - q.lo) + q.lo;
    }
    float freq = q.scale;
    if (MODE == N_SWISS) {                                       // :291-335
        p = domain_warp(p, p, q.warp);
        v3 dsum = V3(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < q.octaves; i++) {
            v3 u = add_cross(p, dsum);
            u = u / dot(u, u);
            v3 p_ = rotate(2.0f * 0.1f * length(dsum), u, p);
            float n = snoise(p_ * freq + so, &tmp);
            dsum = dsum + ((tmp - radial_of(tmp, p)) * -n) * cur;
            fc += (1.0f - fabsf(n)) * cur;
            freq *= q.lacunarity;
            total += cur;
            cur *= q.persistence * smoothstep(-1.0f, 1.0f, fc * fc);
        }
        fc /= total;
        return fc * (q.hi - q.lo) + q.lo;
    }
    if (MODE == N_JORDAN) {                                      // :338-386
        total = cur;
        p = domain_warp(p, p + so, q.warp);
        float n = snoise(p * freq + so, &tmp);
        fc = n * n * cur;
        tmp = tmp * n;
        v3 t0 = tangential(tmp, p);
        v3 dsum_warp = t0 * 0.4f, dsum_damp = t0 * 1.0f;
        float damped = cur * q.persistence;
        for (int i = 1; i < q.octaves; i++) {
            v3 u = add_cross(p, dsum_warp);
            u = u / dot(u, u);
            v3 p_ = rotate(2.0f * 0.1f * length(dsum_warp), u, p);
            n = snoise(p_ * freq + so, &tmp);
            fc += damped * n * n;
            tmp = tmp * n;
            v3 t = tangential(tmp, p);
            dsum_warp = dsum_warp + t * 0.35f;
            dsum_damp = dsum_damp + t * 0.8f;
            freq *= q.lacunarity;
            total += cur;
            cur *= q.persistence;
            damped = cur * (1.0f - 1.0f / one_plus_dot(dsum_damp));
        }
        fc /= total;
        return fc * (q.hi - q.lo) + q.lo;
    }
    // Plateaus :389-437 (r_offset, :419-420, is computed by the reference and never used)
    p = domain_warp(p, p, q.warp);
    for (int i = 0; i < q.octaves; i++) {
        v3 off = so * (float)(i + 1);
        float n = snoise(p * freq + off, &tmp);
        v3 dsum = (((tmp - radial_of(tmp, p)) * (1.0f - fabsf(n))) * n) * 2.0f;
        v3 u = add_cross(p, dsum);
        u = u / dot(u, u);
        v3 p_ = rotate(2.0f * 0.1f * length(dsum), u, p);
        n = snoise(p_ * freq + off, &tmp);
        fc += n * cur / (1.0f + fabsf(fc) * (fabsf(fc) * 5.0f));
        freq *= q.lacunarity;
        total += cur;
        cur *= q.persistence;
    }
    fc /= total;
    return (fc + 1.0f) * 0.5f * (q.hi - q.lo) + q.lo;
}

// filter::blendMode() (Filter.cpp:173-207)
__device__ __forceinline__ float blend_mode(int mode, float old, float nw, float sel)
{
    switch (mode) {
    case DMG_BLEND_REPLACE: return old * (1.0f - sel) + nw * sel;
    case DMG_BLEND_ADD: return old + sel * nw;
    case DMG_BLEND_SUBTRACT: return fmaxf(old - nw, 0.0f);
    case DMG_BLEND_MULTIPLY: return old * nw;
    case DMG_BLEND_DIVIDE: return old / nw;
    case DMG_BLEND_MAX: return fmaxf(old, nw);
    default: return fminf(old, nw);
    }
}

// rows [row0, row0 + nrows) of the canvas; old / sel / out are nrows x W (old == nullptr: 0, sel == nullptr: 1)
template <int MODE>
__global__ void __launch_bounds__(256) k_gradient_noise(canvas c, noise_args q, int row0, const float *old, const float *sel, float *out)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (x >= c.W) return;
    float sx = ((float)x + 0.5f) / (float)c.W, sy = ((float)(row0 + r) + 0.5f) / (float)c.H;
    float lon = sx * (c.c3 - c.c2) + c.c2, lat = sy * (c.c1 - c.c0) + c.c0;        // tex_to_spheric, Shader.h:49-53
    float fc = noise_value<MODE>(q, lon, lat);
    size_t k = (size_t)r * c.W + x;
    out[k] = blend_mode(q.blend, old ? old[k] : 0.0f, fc, sel ? sel[k] : 1.0f);
}

// std::mt19937 (MT19937, init_genrand seeding) + libstdc++'s uniform_real_distribution<float>(0, 10000)
// (GradientNoise.cpp:489-491): generate_canonical<float,24> = float(draw)/2^32, kept below 1, times (b-a) plus a.
// Restated rather than called so the offsets do not depend on the C++ library the host happens to link.
void seed_offset(int seed, float out[3])
{
    uint32_t mt[624];
    mt[0] = (uint32_t)seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (int k = 0; k < 3; k++) {            // the first three outputs only need mt[0..3] and mt[397..399] twisted
        uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        uint32_t v = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11; v ^= (v << 7) & 0x9d2c5680u; v ^= (v << 15) & 0xefc60000u; v ^= v >> 18;
        float r = (float)v / 4294967296.0f;
        if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
        out[k] = r * (10000.0f - 0.0f) + 0.0f;
    }
}

int check_params(const dmg_noise_params *p)
{
    DMG_CHECK_ARG(p != nullptr);
    DMG_CHECK_ARG(p->mode >= DMG_NOISE_DEFAULT && p->mode <= DMG_NOISE_PLATEAUS);
    DMG_CHECK_ARG(p->blend >= DMG_BLEND_REPLACE && p->blend <= DMG_BLEND_MIN);
    DMG_CHECK_ARG(p->octaves >= 0 && p->octaves <= 64);                          // the menu's range, :42
    return DMG_OK;
}

}  // namespace

// device-resident generator: rows [row0, row0+nrows) of `grid`; d_old / d_sel may be null
int noise_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, const dmg_noise_params &p, int row0, int nrows,
                 const float *d_old, const float *d_sel, float *d_out)
{
    canvas c = make_canvas(grid, topo);
    noise_args q;
    q.scale = p.scale; q.lacunarity = p.lacunarity; q.persistence = p.persistence; q.lo = p.min; q.hi = p.max; q.warp = p.warp;
    q.octaves = p.octaves; q.blend = p.blend;
    seed_offset(p.seed, q.so);
    dim3 g((unsigned)((grid.W + 255) / 256), (unsigned)nrows), b(256);
    hipStream_t s = ctx->stream;
    switch (p.mode) {
    case DMG_NOISE_DEFAULT: hipLaunchKernelGGL(k_gradient_noise<N_DEFAULT>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    case DMG_NOISE_RIDGED: hipLaunchKernelGGL(k_gradient_noise<N_RIDGED>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    case DMG_NOISE_BILLOWY: hipLaunchKernelGGL(k_gradient_noise<N_BILLOWY>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    case DMG_NOISE_IQ: hipLaunchKernelGGL(k_gradient_noise<N_IQ>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    case DMG_NOISE_SWISS: hipLaunchKernelGGL(k_gradient_noise<N_SWISS>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    case DMG_NOISE_JORDAN: hipLaunchKernelGGL(k_gradient_noise<N_JORDAN>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    default: hipLaunchKernelGGL(k_gradient_noise<N_PLATEAUS>, g, b, 0, s, c, q, row0, d_old, d_sel, d_out); break;
    }
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

extern "C" {

void dmg_noise_params_default(dmg_noise_params *p)
{
    if (!p) return;
    // GradientNoiseMenu::params, GradientNoise.h:39: {1.0, 2, 0.5, -10, 10, 8, 0, 0.0f, DEFAULT}; blend "Replace" (Filter.cpp:201)
    p->scale = 1.0f; p->lacunarity = 2.0f; p->persistence = 0.5f; p->min = -10.0f; p->max = 10.0f;
    p->octaves = 8; p->seed = 0; p->warp = 0.0f; p->mode = DMG_NOISE_DEFAULT; p->blend = DMG_BLEND_REPLACE;
}

int dmg_noise_seed_offset(int seed, float out3[3])
{
    DMG_CHECK_ARG(out3 != nullptr);
    seed_offset(seed, out3);
    return DMG_OK;
}

int dmg_gradient_noise(dmg_ctx *ctx, const dmg_grid *grid, float *field_inout, const float *sel_or_null,
                       const dmg_noise_params *params)
{
    DMG_CHECK_ARG(ctx != nullptr && field_inout != nullptr);
    int rc = check_params(params);
    if (rc) return rc;
    dmg_topology topo;
    if ((rc = dmg_check_grid(grid, &topo))) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)grid->W * grid->H;
    dev_buf old, sel, out;
    DMG_ALLOC(old, n * 4); DMG_ALLOC(out, n * 4);
    DMG_HIP(hipMemcpyAsync(old.p, field_inout, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (sel_or_null) {
        DMG_ALLOC(sel, n * 4);
        DMG_HIP(hipMemcpyAsync(sel.p, sel_or_null, n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = noise_device(ctx, *grid, topo, *params, 0, grid->H, old.f(), sel_or_null ? sel.f() : nullptr, out.f()))) return rc;
    DMG_HIP(hipMemcpyAsync(field_inout, out.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

int dmg_ocean_generate_terrain(dmg_ocean *o, const dmg_noise_params *params)
{
    DMG_CHECK_ARG(o != nullptr);
    int rc = check_params(params);
    if (rc) return rc;
    DMG_CHECK_ARG(params->blend == DMG_BLEND_REPLACE);        // there is no previous terrain to blend with
    // the band's local rows are grid rows [y0 - gs, y1 + gn)
    if ((rc = noise_device(o->ctx, o->grid, o->topo, *params, o->y0 - o->gs, o->L, nullptr, nullptr, o->img))) return rc;
    return ocean_terrain_changed(o);
}

}  // extern "C"
