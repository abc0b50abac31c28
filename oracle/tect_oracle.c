/* tect_oracle -- CPU restatement of the Tectonics tool's passes (src/filter/tectonics/Tectonics.cpp).
 *
 * TEST INFRASTRUCTURE ONLY (see dmg_oracle.h).  One function per fragment shader; the host loop that strings
 * them together (Tectonics::run, Plate::rotate, the Texture::swap size shuffle) is restated in oracle.py.
 *
 * Textures are RGBA32F, GL_NEAREST, CLAMP_TO_EDGE (src/Texture.cpp:19-36, Plate.cpp:11).  Every texture carries
 * its own size: the scratch maps a, b, c are W x W (Tectonics.cpp:22-24) while plates start W x H.
 *
 * Transcendentals.  The shaders gather with GL_NEAREST through tex -> sphere -> rotate -> sphere -> tex, so the
 * result is a texel choice and any difference in asin/atan/sin/cos shows as a different texel near a boundary.
 * orc_tect_set_math() picks the evaluation:
 *   0  libm (sinf, cosf, asinf, atan2f)
 *   1  asin / atan / atan2 as Mesa 23.2's compiler expands them (the published polynomial forms: asin_expr() of
 *      src/compiler/glsl/builtin_functions.cpp, nir_atan()/nir_atan2() of src/compiler/nir/nir_builtin_builder.c),
 *      libm sin / cos
 *   2  (default) 1 + sin / cos as llvmpipe evaluates them (gallivm lp_build_sincos = the Cephes sinf/cosf
 *      scheme: octant reduction with a three-part pi/4, degree-3 polynomials in z = x*x)
 * Mode 2 is the arithmetic contract of the HIP kernels (bit-exact target); the llvmpipe goldens pin it to a
 * measured texel-mismatch fraction (tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "dmg_oracle.h"

static const float MPI = 3.1415926535897932384626433832795f;
static const float MPI_2 = 1.57079632679489661923f;
static const float MPI_4 = 0.78539816339744830962f;

static int g_math = 2;
void orc_tect_set_math(int mode) { g_math = mode; }

/* ---------- transcendental variants ---------- */
static inline float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* asin_expr(x, 0.086566724, -0.03102955) */
static inline float mesa_asin(float x)
{
    float ax = fabsf(x);
    float t = 0.086566724f + ax * -0.03102955f;
    t = (MPI_4 - 1.0f) + ax * t;
    t = MPI_2 + ax * t;
    return sgn(x) * (MPI_2 - sqrtf(1.0f - ax) * t);
}

/* atan of a non-negative argument: range reduction to [0,1], then the odd polynomial of degree 11 summed term by
 * term from the lowest power (this order reproduces llvmpipe on 99.98 % of random arguments; Horner form: 76 %) */
static inline float mesa_atan_pos(float y_over_x)
{
    float a = fabsf(y_over_x);
    float x = fminf(a, 1.0f) / fmaxf(a, 1.0f);
    float x2 = x * x, x3 = x2 * x, x5 = x3 * x2, x7 = x5 * x2, x9 = x7 * x2;
    float p = x * 0.9999793128310355f;
    p = p + x3 * -0.3326756418091246f;
    p = p + x5 * 0.1938924977115610f;
    p = p + x7 * -0.1173503194786851f;
    p = p + x9 * 0.0536813784310406f;
    p = p + (x9 * -0.0121323213173444f) * x2;   /* x^11 is used once: Mesa multiplies x^9 by the constant first (probe: 131072/131072) */
    p = p + (a > 1.0f ? 1.0f : 0.0f) * (p * -2.0f + MPI_2);
    return p * sgn(y_over_x);
}

/* _atan2(y, x) */
static inline float mesa_atan2(float y, float x)
{
    int flip = 0.0f >= x;
    float s = flip ? fabsf(x) : y;
    float t = flip ? y : fabsf(x);
    float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
    float rcp_scaled_t = 1.0f / (t * scale);
    float s_over_t = (s * scale) * rcp_scaled_t;
    float tn = fabsf(x) == fabsf(y) ? 1.0f : fabsf(s_over_t);
    float arc = (flip ? 1.0f : 0.0f) * MPI_2 + mesa_atan_pos(tn);
    return fminf(y, rcp_scaled_t) < 0.0f ? -arc : arc;
}

/* Cephes sinf/cosf as gallivm's lp_build_sincos strings it together */
static inline float lp_sincos(float xin, int want_cos)
{
    const float FOPI = 1.27323954473516f;
    const float DP1 = -0.78515625f, DP2 = -2.4187564849853515625e-4f, DP3 = -3.77489497744594108e-8f;
    uint32_t xbits;
    memcpy(&xbits, &xin, 4);
    uint32_t sign_bit = want_cos ? 0u : (xbits & 0x80000000u);
    float x = fabsf(xin);
    float y = x * FOPI;
    int32_t j = (int32_t)y;
    j = (j + 1) & ~1;
    y = (float)j;
    /* sine: octant j; cosine: octant j - 2.  Bit 2 flips the sign, bit 1 picks the cosine polynomial. */
    uint32_t poly_cos;
    if (want_cos) {
        int32_t j2 = j - 2;
        sign_bit = ((uint32_t)(~j2 & 4)) << 29;
        poly_cos = (j2 & 2) != 0;
    } else {
        sign_bit ^= ((uint32_t)(j & 4)) << 29;
        poly_cos = (j & 2) != 0;
    }
    /* gallivm emits llvm.fmuladd at these sites; on an FMA host they are fused (verified against llvmpipe:
     * fused = 0 mismatches in 16384 random arguments, unfused = 140) */
    x = fmaf(y, DP1, x);
    x = fmaf(y, DP2, x);
    x = fmaf(y, DP3, x);
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
    float r = poly_cos ? yc : ys;
    uint32_t rb;
    memcpy(&rb, &r, 4);
    rb ^= sign_bit;
    memcpy(&r, &rb, 4);
    return r;
}

static inline float t_sin(float x) { return g_math >= 2 ? lp_sincos(x, 0) : sinf(x); }
static inline float t_cos(float x) { return g_math >= 2 ? lp_sincos(x, 1) : cosf(x); }
static inline float t_asin(float x) { return g_math >= 1 ? mesa_asin(x) : asinf(x); }
static inline float t_atan2(float y, float x) { return g_math >= 1 ? mesa_atan2(y, x) : atan2f(y, x); }

/* ---------- texture unit ---------- */
typedef struct { float x, y, z, w; } vec4;
static inline vec4 v4(float x, float y, float z, float w) { vec4 r = { x, y, z, w }; return r; }
static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

static inline vec4 fetch4(const float *t, int w, int h, float u, float v)
{
    int i = clampi((int)floorf(u * (float)w), 0, w - 1);
    int j = clampi((int)floorf(v * (float)h), 0, h - 1);
    const float *p = t + ((size_t)j * w + i) * 4;
    return v4(p[0], p[1], p[2], p[3]);
}
static inline void store4(float *t, int w, int x, int y, vec4 c)
{
    float *p = t + ((size_t)y * w + x) * 4;
    p[0] = c.x; p[1] = c.y; p[2] = c.z; p[3] = c.w;
}

/* ---------- geometry (src/Shader.h:49-67) with the transcendental variant in force ---------- */
static inline void to_spheric(const orc_grid *g, float sx, float sy, float *lon, float *lat)
{ orc_tex_to_spheric(g, sx, sy, lon, lat); }

static inline void to_tex(const orc_grid *g, float lon, float lat, float *sx, float *sy)
{   /* spheric_to_tex :55-59 */
    *sx = (lon - g->coords[2]) / (g->coords[3] - g->coords[2]);
    *sy = (lat - g->coords[0]) / (g->coords[1] - g->coords[0]);
}

/* plateTexture / inverseplateTexture (Tectonics.cpp:70-92): m is the mat3 uniform, column-major */
static inline vec4 rotated_fetch(const orc_grid *g, const float *t, int w, int h, const float *m, float sx, float sy)
{
    float lon, lat;
    to_spheric(g, sx, sy, &lon, &lat);
    float cx = t_cos(lat) * t_cos(lon), cy = t_cos(lat) * t_sin(lon), cz = t_sin(lat);   /* :61-63 */
    float rx = m[0] * cx + m[3] * cy + m[6] * cz;
    float ry = m[1] * cx + m[4] * cy + m[7] * cz;
    float rz = m[2] * cx + m[5] * cy + m[8] * cz;
    float lon2 = t_atan2(ry, rx), lat2 = t_asin(rz);                                      /* :65-67 */
    float u, v;
    to_tex(g, lon2, lat2, &u, &v);
    return fetch4(t, w, h, u, v);
}

/* geodistance (src/Shader.h:349-354); pow(x, 2) is x*x after Mesa's algebraic pass */
static inline float geodistance(const orc_grid *g, float p1x, float p1y, float p2x, float p2y, float sizex)
{
    float l1, a1, l2, a2;
    to_spheric(g, p1x, p1y, &l1, &a1);
    to_spheric(g, p2x, p2y, &l2, &a2);
    float s1 = t_sin(fabsf(a2 - a1) / 2.0f);
    float s2 = t_sin((l1 - l2) / 2.0f);
    float ds = 2.0f * t_asin(sqrtf(s1 * s1 + t_cos(a1) * t_cos(a2) * (s2 * s2)));
    return ds / (g->coords[3] - g->coords[2]) * sizex;
}

#define ST(x, n) (((float)(x) + 0.5f) / (float)(n))
/* st of the fragment (x, y) of a w x h target: (x+0.5)/w, or as llvmpipe's rasteriser interpolates it (dmg_oracle.c,
 * orc_set_raster_st: identical on power-of-two targets) */
int orc_raster_st(void);
void orc_st_raster_wh(int W, int H, int x, int y, float *sx, float *sy);
static inline void st_of(int x, int y, int w, int h, float *sx, float *sy)
{
    *sx = ST(x, w); *sy = ST(y, h);
    if (orc_raster_st()) orc_st_raster_wh(w, h, x, y, sx, sy);
}

/* ---------- passes ---------- */
/* plate initialisation, Tectonics.cpp:28-34 (which = 0) and :44-50 (which = 1); img and dst are W x H */
void orc_tect_init_plate(const orc_grid *g, const float *img, int which, float *dst)
{
    for (int y = 0; y < g->H; y++)
        for (int x = 0; x < g->W; x++) {
            float a = img[(size_t)y * g->W + x], sx, sy_unused;
            st_of(x, y, g->W, g->H, &sx, &sy_unused);
            vec4 fc;
            if (which == 0) {
                float h = a > 0.0f ? a : -1.0f;
                fc = sx > 0.5f ? v4(h, 1.0f, 0, 0) : v4(0, -1.0f, 0, 0);
            } else {
                float h = a > 0.0f ? a : -2.0f;
                fc = sx < 0.5f ? v4(h, 0.5f, 0, 0) : v4(0, -1.0f, 0, 0);
            }
            store4(dst, g->W, x, y, fc);
        }
}

/* setzero :99-106 */
void orc_tect_fold_zero(float *dst, int w, int h)
{
    for (size_t i = 0; i < (size_t)w * h; i++) {
        dst[4 * i] = 0.0f; dst[4 * i + 1] = -1.0f; dst[4 * i + 2] = -1.0f; dst[4 * i + 3] = -1.1e6f;
    }
}

/* fold :112-149: dst (w x h) = foldtex overlaid with one rotated plate */
void orc_tect_fold(const orc_grid *g, const float *foldtex, int fw, int fh, const float *plate, int pw, int ph,
                   const float *rot, float index, float *dst, int w, int h)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sx, sy;
            st_of(x, y, w, h, &sx, &sy);
            vec4 fc = fetch4(foldtex, fw, fh, sx, sy);
            vec4 p = rotated_fetch(g, plate, pw, ph, rot, sx, sy);
            float plateHeight = p.x, plateAge = p.y;
            float previousAge = fc.z, previousHeight = fc.y;
            int overlap = plateAge >= 0.0f && previousAge >= 0.0f;
            if (overlap) fc.w = plateHeight;
            if (plateAge >= 0.0f && !overlap) fc = v4(index, plateHeight, plateAge, fc.w);
            int landOnOcean = plateHeight > 0.0f && previousHeight <= 0.0f;
            int plateYounger = plateAge < previousAge;
            int plateOnTop = (plateYounger && plateHeight <= 0.0f && previousHeight <= 0.0f) ||
                             (!plateYounger && plateHeight > 0.0f && previousHeight > 0.0f) || landOnOcean;
            if (overlap && plateOnTop) fc = v4(index, plateHeight, plateAge, previousHeight);
            store4(dst, w, x, y, fc);
        }
}

/* oceanSpreading, first and last pass: :303-314 (mode 0: z = 0) and :376-387 (mode 1: z = z>0 ? 1 : 0) */
void orc_tect_spread_flag(const float *src, int mode, float *dst, int w, int h)
{
    for (size_t i = 0; i < (size_t)w * h; i++) {
        dst[4 * i] = src[4 * i]; dst[4 * i + 1] = src[4 * i + 1]; dst[4 * i + 3] = src[4 * i + 3];
        dst[4 * i + 2] = mode == 0 ? 0.0f : (src[4 * i + 2] > 0.0f ? 1.0f : 0.0f);
    }
}

/* oceanSpreading ring pass :325-350.  The 16-tap loop is NOT unrolled by Mesa (verified: folding cos/sin of the tap
 * angles with libm breaks ties differently from llvmpipe on a handful of texels per pass; evaluating them with the
 * runtime sincos reproduces it), so the angles go through the transcendental variant in force. */
void orc_tect_spread_ring(const orc_grid *g, const float *src, float radius, float *dst, int w, int h)
{
    float ct[16], sn[16];
    for (int i = 0; i < 16; i++) {
        float ang = 2.0f * 3.14159f * (float)i / 16.0f;
        ct[i] = t_cos(ang);
        sn[i] = t_sin(ang);
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sx, sy;
            st_of(x, y, w, h, &sx, &sy);
            vec4 fc = fetch4(src, w, h, sx, sy);
            float lon, phi;
            to_spheric(g, sx, sy, &lon, &phi);
            float factor = 1.0f / t_cos(fabsf(phi));
            for (int i = 0; i < 16; i++) {
                float nx, ny;
                orc_offset(g, sx, sy, ct[i] * radius * factor, sn[i] * radius, (float)w, (float)h, &nx, &ny);
                vec4 a = fetch4(src, w, h, nx, ny);
                float nz = a.z + geodistance(g, nx, ny, sx, sy, (float)w);
                if ((nz < fc.z || fc.x == 0.0f) && a.x != 0.0f) fc = v4(a.x, -1.1f, nz, -1.1e6f);
            }
            store4(dst, w, x, y, fc);
        }
}

/* render :186-200: terrain (W x H, R32F) = foldtex.y */
void orc_tect_render(const orc_grid *g, const float *foldtex, int fw, int fh, float *terrain)
{
    for (int y = 0; y < g->H; y++)
        for (int x = 0; x < g->W; x++)
        {
            float sx, sy;
            st_of(x, y, g->W, g->H, &sx, &sy);
            terrain[(size_t)y * g->W + x] = fetch4(foldtex, fw, fh, sx, sy).y;
        }
}

/* the only channel of collision() (:399-618) that anything reads afterwards: c.y = plateIndices.x (:590) */
void orc_tect_collision_y(const float *plateIndices, int bw, int bh, float *c, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            float sx, sy;
            st_of(x, y, w, h, &sx, &sy);
            c[((size_t)y * w + x) * 4 + 1] = fetch4(plateIndices, bw, bh, sx, sy).x;
        }
}

/* back to plates :216-251: dst (w x h) from the global map `foldtex` and the plate's previous texture */
void orc_tect_unfold(const orc_grid *g, const float *foldtex, int fw, int fh, const float *oldPlate, int ow, int oh,
                     const float *invrot, float index, float *dst, int w, int h)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sx, sy;
            st_of(x, y, w, h, &sx, &sy);
            vec4 a = rotated_fetch(g, foldtex, fw, fh, invrot, sx, sy);
            vec4 fc = fetch4(oldPlate, ow, oh, sx, sy);
            if (fc.y >= 0.0f) fc.y += 0.01f;
            float lon, phi;
            to_spheric(g, sx, sy, &lon, &phi);
            float factor = 1.0f / t_cos(fabsf(phi));
            int differentIndex = 1;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++) {
                    float ox, oy;
                    orc_offset(g, sx, sy, (float)i * factor, (float)j, (float)fw, (float)fh, &ox, &oy);
                    vec4 n = rotated_fetch(g, foldtex, fw, fh, invrot, ox, oy);
                    differentIndex = differentIndex && n.x != index;
                }
            if (differentIndex && !(a.y <= 0.0f && fc.x > 0.0f)) fc = v4(0, -1.0f, 0, 0);
            if (fc.y < 0.0f && a.x == index && fabsf(a.z - 1.0f) < 0.01f) fc = v4(-index, 1.0f, 0, 0);
            store4(dst, w, x, y, fc);
        }
}

/* the transcendental variant in force, sampled: out = (sin x, cos x, asin y, atan(z, w)) per RGBA input */
void orc_tect_math_eval(int n, const float *in4, float *out4)
{
    for (int i = 0; i < n; i++) {
        const float *a = in4 + 4 * i;
        out4[4 * i] = t_sin(a[0]); out4[4 * i + 1] = t_cos(a[0]);
        out4[4 * i + 2] = t_asin(a[1]); out4[4 * i + 3] = t_atan2(a[2], a[3]);
    }
}
