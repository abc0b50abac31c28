/* noise_oracle -- CPU restatement of the GradientNoise tool (src/filter/GradientNoise.cpp).
 *
 * TEST INFRASTRUCTURE ONLY (see dmg_oracle.h).  Statement-by-statement fp32 translation of the GLSL: the 3-D simplex
 * noise with analytic gradient the reference embeds (:66-175, Ashima Arts / stegu webgl-noise, MIT), the seven octave
 * loops (:181-437), the blend functions of filter::blendMode() (Filter.cpp:170-207) and the seed offsets (:489-491:
 * std::mt19937 + libstdc++'s uniform_real_distribution<float>, restated from their published algorithms).
 *
 * sin/cos use the evaluation pinned for Tectonics (tect_oracle.c, mode 2 = gallivm's): the octave loop adds seed
 * offsets of up to 10^4 to a unit-sphere position, so an ulp of difference in the position can move the sum by 1e-3.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "dmg_oracle.h"
int orc_raster_st(void);
void orc_st_raster(const orc_grid *g, int x, int y, float *sx, float *sy);

static const float MPI = 3.1415926535897932384626433832795f;

/* gallivm sin/cos (same scheme as tect_oracle.c; duplicated so each restatement stands alone) */
static inline void n_sincos(float xin, float *sn, float *cs)
{
    float x = fabsf(xin);
    int32_t j = (int32_t)(x * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    x = fmaf(y, -0.78515625f, x);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = fmaf(yc, z, -1.388731625493765E-003f);
    yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = yc * z;
    yc = yc - z * 0.5f;
    yc = yc + 1.0f;
    float ys = -1.9515295891E-4f;
    ys = fmaf(ys, z, 8.3321608736E-3f);
    ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    int32_t j2 = j - 2;
    float s = (j & 2) ? yc : ys, c = (j2 & 2) ? yc : ys;
    uint32_t sb, cb, xb;
    memcpy(&xb, &xin, 4); memcpy(&sb, &s, 4); memcpy(&cb, &c, 4);
    sb ^= (xb & 0x80000000u) ^ (((uint32_t)(j & 4)) << 29);
    cb ^= ((uint32_t)(~j2 & 4)) << 29;
    memcpy(sn, &sb, 4); memcpy(cs, &cb, 4);
}
static inline float n_sin(float x) { float s, c; n_sincos(x, &s, &c); return s; }
static inline float n_cos(float x) { float s, c; n_sincos(x, &s, &c); return c; }

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 adds3(v3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }
static inline v3 mul3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
/* Mesa lowers dot() from the last component down: (a.z*b.z + a.y*b.y) + a.x*b.x (pinned against probe shaders) */
static inline float dot3(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
/* dot(a, vec3(c)): the products share a factor and NIR's algebraic pass pulls it out: ((a.z + a.y) + a.x) * c */
static inline float dot3c(v3 a, float c) { return ((a.z + a.y) + a.x) * c; }
/* 1 + dot(a,a) where the dot has no other use: NIR reassociates the constant to the innermost add (probe-pinned) */
static inline float one_plus_dot(v3 a) { return ((1.0f + a.z * a.z) + a.y * a.y) + a.x * a.x; }
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
/* p + cross(p, d): NIR reassociates the single-use difference into (p + a*b) - c*d per component (probe-pinned) */
static inline v3 add_cross(v3 p, v3 d)
{ return V3((p.x + p.y * d.z) - p.z * d.y, (p.y + p.z * d.x) - p.x * d.z, (p.z + p.x * d.y) - p.y * d.x); }
static inline float mod289f(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
static inline float permute(float x) { return mod289f(((x * 34.0f) + 1.0f) * x); }
static inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float signf_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* snoise(v, out gradient) :97-175 */
static float snoise(v3 v, v3 *gradient)
{
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    float s = dot3c(v, Cy);
    v3 i = V3(floorf(v.x + s), floorf(v.y + s), floorf(v.z + s));
    float t = dot3c(i, Cx);
    v3 x0 = adds3(sub3(v, i), t);
    v3 g = V3(stepf(x0.y, x0.x), stepf(x0.z, x0.y), stepf(x0.x, x0.z));
    v3 l = V3(1.0f - g.x, 1.0f - g.y, 1.0f - g.z);
    v3 i1 = V3(fminf(g.x, l.z), fminf(g.y, l.x), fminf(g.z, l.y));
    v3 i2 = V3(fmaxf(g.x, l.z), fmaxf(g.y, l.x), fmaxf(g.z, l.y));
    v3 x1 = adds3(sub3(x0, i1), Cx);
    v3 x2 = adds3(sub3(x0, i2), Cy);
    v3 x3 = V3(x0.x - 0.5f, x0.y - 0.5f, x0.z - 0.5f);
    i = V3(mod289f(i.x), mod289f(i.y), mod289f(i.z));
    float oz[4] = { 0.0f, i1.z, i2.z, 1.0f }, oy[4] = { 0.0f, i1.y, i2.y, 1.0f }, ox[4] = { 0.0f, i1.x, i2.x, 1.0f };
    float p[4];
    for (int k = 0; k < 4; k++) {
        float a = permute(i.z + oz[k]);
        float b = permute((a + i.y) + oy[k]);
        p[k] = permute((b + i.x) + ox[k]);
    }
    const float n_ = 0.142857142857f;
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
    float X[4], Y[4], Hh[4];
    for (int k = 0; k < 4; k++) {
        float j = p[k] - 49.0f * floorf(p[k] * nsz * nsz);
        float x_ = floorf(j * nsz);
        float y_ = floorf(j - 7.0f * x_);
        X[k] = x_ * nsx + nsy;
        Y[k] = y_ * nsx + nsy;
        Hh[k] = 1.0f - fabsf(X[k]) - fabsf(Y[k]);
    }
    v3 P[4];
    for (int k = 0; k < 4; k++) {
        float sh = -stepf(Hh[k], 0.0f);
        float ax = X[k] + (floorf(X[k]) * 2.0f + 1.0f) * sh;
        float ay = Y[k] + (floorf(Y[k]) * 2.0f + 1.0f) * sh;
        P[k] = V3(ax, ay, Hh[k]);
    }
    for (int k = 0; k < 4; k++) {
        float norm = 1.79284291400159f - 0.85373472095314f * dot3(P[k], P[k]);
        P[k] = mul3(P[k], norm);
    }
    v3 xs[4] = { x0, x1, x2, x3 };
    float m[4], m2[4], m4[4], pdotx[4], temp[4];
    for (int k = 0; k < 4; k++) {
        m[k] = fmaxf(0.6f - dot3(xs[k], xs[k]), 0.0f);
        m2[k] = m[k] * m[k];
        m4[k] = m2[k] * m2[k];
        pdotx[k] = dot3(P[k], xs[k]);
        temp[k] = m2[k] * m[k] * pdotx[k];
    }
    v3 gr = add3(add3(add3(mul3(x0, temp[0]), mul3(x1, temp[1])), mul3(x2, temp[2])), mul3(x3, temp[3]));
    gr = mul3(gr, -8.0f);
    v3 gp = add3(add3(add3(mul3(P[0], m4[0]), mul3(P[1], m4[1])), mul3(P[2], m4[2])), mul3(P[3], m4[3]));
    gr = add3(gr, gp);
    *gradient = mul3(gr, 42.0f);
    return 42.0f * (((m4[3] * pdotx[3] + m4[2] * pdotx[2]) + m4[1] * pdotx[1]) + m4[0] * pdotx[0]);
}

/* rotation_matrix(theta, u) * p (src/Shader.h:35-41) */
static v3 rotate(float theta, v3 u, v3 p)
{
    float c = n_cos(theta), s = n_sin(theta), oc = 1.0f - c;
    v3 c0 = V3(c + u.x * u.x * oc, u.y * u.x * oc + u.z * s, u.z * u.x * oc - u.y * s);
    v3 c1 = V3(u.x * u.y * oc - u.z * s, c + u.y * u.y * oc, u.z * u.y * oc + u.x * s);
    v3 c2 = V3(u.x * u.z * oc + u.y * s, u.y * u.z * oc - u.x * s, c + u.z * u.z * oc);
    return add3(add3(mul3(c0, p.x), mul3(c1, p.y)), mul3(c2, p.z));
}

/* tmp - dot(tmp,p)/length(p) * p/length(p) */
static inline v3 radial_of(v3 tmp, v3 p) { return div3(mul3(p, dot3(tmp, p) / len3(p)), len3(p)); }
static inline v3 tangential(v3 tmp, v3 p) { return sub3(tmp, radial_of(tmp, p)); }

/* the "domain warp" prologue shared by DEFAULT/SWISS/Jordan/Plateaus */
static v3 domain_warp(v3 p, v3 at, float warp)
{
    v3 tmp;
    snoise(at, &tmp);
    tmp = tangential(tmp, p);
    v3 u = add3(p, tmp);
    u = div3(u, dot3(u, u));
    float theta = warp * 0.1f * len3(tmp);
    return rotate(theta, u, p);
}

static inline float smoothstepf(float e0, float e1, float x)
{
    float t = (x - e0) / (e1 - e0);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return t * (t * (3.0f - 2.0f * t));      /* Mesa's association (pinned against a probe shader) */
}

typedef struct {
    float scale, lacunarity, persistence, lo, hi, warp;
    int octaves, mode;
    float seed_offset[3];
} noise_prm;

static float noise_value(const orc_grid *g, const noise_prm *q, float sx, float sy)
{
    float lon, lat;
    orc_tex_to_spheric(g, sx, sy, &lon, &lat);
    v3 so = V3(q->seed_offset[0], q->seed_offset[1], q->seed_offset[2]);
    float cur = 1.0f, total = 0.0f, fc = 0.0f;
    v3 tmp;
    v3 p;
    if (q->mode == 0) p = mul3(V3(n_cos(lat) * n_cos(lon), n_cos(lat) * n_sin(lon), n_sin(lat)), q->scale);   /* :183 */
    else {   /* vec3(sin(M_PI/2-geo.y)*cos(geo.x), sin(M_PI/2-geo.y)*sin(geo.x), cos(M_PI/2-geo.y)) :227 */
        float co = MPI / 2.0f - lat;
        p = V3(n_sin(co) * n_cos(lon), n_sin(co) * n_sin(lon), n_cos(co));
        if (q->mode >= 1 && q->mode <= 3) p = mul3(p, q->scale);
    }
    switch (q->mode) {
    case 0: /* DEFAULT :182-221 */
        p = domain_warp(p, p, q->warp);
        for (int i = 0; i < q->octaves; i++) {
            fc += snoise(add3(p, mul3(so, (float)(i + 1))), &tmp) * cur;
            p = mul3(p, q->lacunarity);
            total += cur;
            cur *= q->persistence;
        }
        fc /= total;
        return (fc + 1.0f) * 0.5f * (q->hi - q->lo) + q->lo;
    case 1: /* RIDGED :224-242 */
    case 2: /* BILLOWY :245-263 */
        for (int i = 0; i < q->octaves; i++) {
            float n = fabsf(snoise(add3(p, so), &tmp));
            fc += (q->mode == 1 ? 1.0f - n : n) * cur;
            p = mul3(p, q->lacunarity);
            total += cur;
            cur *= q->persistence;
        }
        fc /= total;
        return fc * (q->hi - q->lo) + q->lo;
    case 3: { /* IQNoise :266-288 */
        v3 dsum = V3(0, 0, 0);
        for (int i = 0; i < q->octaves; i++) {
            float n = snoise(add3(p, so), &tmp) * cur;
            v3 radial = radial_of(tmp, p);
            dsum = add3(dsum, sub3(tmp, radial));
            fc += n / (1.0f + dot3(dsum, dsum));
            p = mul3(p, q->lacunarity);
            total += cur / (1.0f + dot3(dsum, dsum));
            cur *= q->persistence;
        }
        fc /= total;
        return (fc + 1.0f) * 0.5f * (q->hi - q->lo) + q->lo;
    }
    case 4: { /* SWISS :291-335 */
        float freq = q->scale;
        p = domain_warp(p, p, q->warp);
        v3 dsum = V3(0, 0, 0);
        for (int i = 0; i < q->octaves; i++) {
            v3 u = add_cross(p, dsum);
            u = div3(u, dot3(u, u));
            float theta = 2.0f * 0.1f * len3(dsum);
            v3 p_ = rotate(theta, u, p);
            float n = snoise(add3(mul3(p_, freq), so), &tmp);
            v3 radial = radial_of(tmp, p);
            dsum = add3(dsum, mul3(mul3(sub3(tmp, radial), -n), cur));
            fc += (1.0f - fabsf(n)) * cur;
            freq *= q->lacunarity;
            total += cur;
            cur *= q->persistence * smoothstepf(-1.0f, 1.0f, fc * fc);
        }
        fc /= total;
        return fc * (q->hi - q->lo) + q->lo;
    }
    case 5: { /* Jordan :338-386 */
        float freq = q->scale;
        total = cur;
        p = domain_warp(p, add3(p, so), q->warp);
        float n = snoise(add3(mul3(p, freq), so), &tmp);
        fc = n * n * cur;
        tmp = mul3(tmp, n);
        v3 dsum_warp = mul3(tangential(tmp, p), 0.4f);
        v3 dsum_damp = mul3(tangential(tmp, p), 1.0f);
        float damped = cur * q->persistence;
        for (int i = 1; i < q->octaves; i++) {
            v3 u = add_cross(p, dsum_warp);
            u = div3(u, dot3(u, u));
            float theta = 2.0f * 0.1f * len3(dsum_warp);
            v3 p_ = rotate(theta, u, p);
            n = snoise(add3(mul3(p_, freq), so), &tmp);
            fc += damped * n * n;
            tmp = mul3(tmp, n);
            dsum_warp = add3(dsum_warp, mul3(tangential(tmp, p), 0.35f));
            dsum_damp = add3(dsum_damp, mul3(tangential(tmp, p), 0.8f));
            freq *= q->lacunarity;
            total += cur;
            cur *= q->persistence;
            damped = cur * (1.0f - 1.0f / one_plus_dot(dsum_damp));
        }
        fc /= total;
        return fc * (q->hi - q->lo) + q->lo;
    }
    default: { /* Plateaus :389-437 */
        float freq = q->scale;
        p = domain_warp(p, p, q->warp);
        for (int i = 0; i < q->octaves; i++) {
            v3 off = mul3(so, (float)(i + 1));
            float n = snoise(add3(mul3(p, freq), off), &tmp);
            v3 radial = radial_of(tmp, p);
            v3 dsum = mul3(mul3(mul3(sub3(tmp, radial), 1.0f - fabsf(n)), n), 2.0f);
            v3 u = add_cross(p, dsum);
            u = div3(u, dot3(u, u));
            float theta = 2.0f * 0.1f * len3(dsum);
            v3 p_ = rotate(theta, u, p);
            n = snoise(add3(mul3(p_, freq), off), &tmp);      /* r_offset (:419-420) is computed and never used */
            fc += n * cur / (1.0f + fabsf(fc) * (fabsf(fc) * 5.0f));   /* Mesa moves the constant inwards (probe-pinned) */
            freq *= q->lacunarity;
            total += cur;
            cur *= q->persistence;
        }
        fc /= total;
        return (fc + 1.0f) * 0.5f * (q->hi - q->lo) + q->lo;
    }
    }
}

/* filter::blendMode() (Filter.cpp:173-207): 0 replace, 1 add, 2 subtract, 3 multiply, 4 divide, 5 max, 6 min */
static inline float blend(int mode, float old, float nw, float sel)
{
    switch (mode) {
    case 0: return old * (1.0f - sel) + nw * sel;
    case 1: return old + sel * nw;
    case 2: return fmaxf(old - nw, 0.0f);
    case 3: return old * nw;
    case 4: return old / nw;
    case 5: return fmaxf(old, nw);
    default: return fminf(old, nw);
    }
}

/* std::mt19937 (MT19937, Matsumoto & Nishimura) seeded with init_genrand(seed), and libstdc++'s
 * uniform_real_distribution<float>(0, 10000): generate_canonical<float, 24> = float(draw) / 2^32, clamped below 1,
 * then * (b - a) + a  (GradientNoise.cpp:489-491) */
void orc_noise_seed_offset(int seed, float *out3)
{
    uint32_t mt[624];
    mt[0] = (uint32_t)seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (int k = 0; k < 3; k++) {
        uint32_t y = mt[k];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        float r = (float)y / 4294967296.0f;
        if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
        out3[k] = r * (10000.0f - 0.0f) + 0.0f;
    }
}

/* GradientNoiseFilter::step :61-500.  prm = {scale, lacunarity, persistence, min, max, warp}, mode = noise_mode enum
 * (GradientNoise.h:10), img / sel / out are W*H. */
void orc_gradient_noise(const orc_grid *g, const float *img, const float *sel, const float *prm6, int octaves, int seed,
                        int mode, int blend_mode, float *out)
{
    noise_prm q;
    q.scale = prm6[0]; q.lacunarity = prm6[1]; q.persistence = prm6[2]; q.lo = prm6[3]; q.hi = prm6[4]; q.warp = prm6[5];
    q.octaves = octaves; q.mode = mode;
    orc_noise_seed_offset(seed, q.seed_offset);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < g->H; y++)
        for (int x = 0; x < g->W; x++) {
            float sx = ((float)x + 0.5f) / (float)g->W, sy = ((float)y + 0.5f) / (float)g->H;
            if (orc_raster_st()) orc_st_raster(g, x, y, &sx, &sy);      /* llvmpipe's interpolated st (dmg_oracle.c) */
            float fc = noise_value(g, &q, sx, sy);
            size_t k = (size_t)y * g->W + x;
            out[k] = blend(blend_mode, img[k], fc, sel ? sel[k] : 1.0f);
        }
}

/* snoise sampled at arbitrary points (value, gradient xyz): for pinning against probe shaders */
void orc_snoise_eval(int n, const float *xyz, float *out4)
{
    for (int i = 0; i < n; i++) {
        v3 gr;
        out4[4 * i] = snoise(V3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), &gr);
        out4[4 * i + 1] = gr.x; out4[4 * i + 2] = gr.y; out4[4 * i + 3] = gr.z;
    }
}
