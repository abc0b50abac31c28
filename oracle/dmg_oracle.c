/* dmg_oracle.c -- see dmg_oracle.h.  TEST INFRASTRUCTURE ONLY (parity checker).
 *
 * Plain C restatement of the reference's GLSL passes.  "Literal" functions walk
 * the shader text statement by statement through restated texture-unit
 * semantics; "rows" functions are the hoisted form (per-row tables + integer
 * neighbour indices) that the HIP kernels implement and must match bit for bit.
 */
#include "dmg_oracle.h"
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* GLSL `#define M_PI 3.1415926535897932384626433832795` is a float literal (src/Shader.h:30-31) */
static const float MPI = 3.1415926535897932384626433832795f;

static int g_bilinear = 0;
static int g_threads = 0;
void orc_set_linear_taps(int bilinear) { g_bilinear = bilinear; }
void orc_set_threads(int n) { g_threads = n; }
/* Default: at most 16 threads (a one-GPU box's CPU share; the host may show far more cores than the cgroup
 * grants, and oversubscribed OpenMP barriers crawl).  orc_set_threads() overrides. */
static int nthreads(void)
{
#ifdef _OPENMP
    if (g_threads > 0) return g_threads;
    int n = omp_get_num_procs();
    return n > 16 ? 16 : (n < 1 ? 1 : n);
#else
    return 1;
#endif
}
/* small grids: no more threads than 16-row chunks */
static int nthreads_rows(int H, int want)
{
    int n = want > 0 ? want : nthreads();
    int cap = H / 16;
    if (cap < 1) cap = 1;
    return n < cap ? n : cap;
}

/* ---------- GLSL transcendentals ----------
 * mode 1 (default): as llvmpipe evaluates them -- asin / acos / atan / atan2 in the polynomial forms Mesa 23.2's
 * compiler expands them to, sin / cos in gallivm's Cephes scheme, tan = sin / cos (pinned with probe shaders, see
 * tect_oracle.c); mode 0: libm.  Expressions of constants are folded by Mesa on the host and stay libm calls. */
static int g_glsl_math = 1;
void orc_set_glsl_math(int mode) { g_glsl_math = mode; }
static inline float m_sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline void lp_sincos(float xin, float *sn, float *cs)
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
static inline float gsin(float x) { if (!g_glsl_math) return sinf(x); float s, c; lp_sincos(x, &s, &c); return s; }
static inline float gcos(float x) { if (!g_glsl_math) return cosf(x); float s, c; lp_sincos(x, &s, &c); return c; }
static inline float gtan(float x) { if (!g_glsl_math) return tanf(x); float s, c; lp_sincos(x, &s, &c); return s / c; }
static inline float m_asin_poly(float x, float p0, float p1)
{
    float ax = fabsf(x);
    float t = p0 + ax * p1;
    t = (0.78539816339744830962f - 1.0f) + ax * t;
    t = 1.57079632679489661923f + ax * t;
    return m_sgn(x) * (1.57079632679489661923f - sqrtf(1.0f - ax) * t);
}
static inline float gasin(float x) { return g_glsl_math ? m_asin_poly(x, 0.086566724f, -0.03102955f) : asinf(x); }
static inline float gacos(float x) { return g_glsl_math ? 1.57079632679489661923f - m_asin_poly(x, 0.08132463f, -0.02363318f) : acosf(x); }
static inline float m_atan_pos(float a)
{
    float x = fminf(a, 1.0f) / fmaxf(a, 1.0f);
    float x2 = x * x, x3 = x2 * x, x5 = x3 * x2, x7 = x5 * x2, x9 = x7 * x2;
    float p = x * 0.9999793128310355f;
    p = p + x3 * -0.3326756418091246f;
    p = p + x5 * 0.1938924977115610f;
    p = p + x7 * -0.1173503194786851f;
    p = p + x9 * 0.0536813784310406f;
    p = p + (x9 * -0.0121323213173444f) * x2;   /* x^11 is used once: Mesa multiplies x^9 by the constant first (probe: 131072/131072) */
    return p + (a > 1.0f ? 1.0f : 0.0f) * (p * -2.0f + 1.57079632679489661923f);
}
static inline float gatan(float x) { return g_glsl_math ? m_atan_pos(fabsf(x)) * m_sgn(x) : atanf(x); }
static inline float gatan2(float y, float x)
{
    if (!g_glsl_math) return atan2f(y, x);
    int flip = 0.0f >= x;
    float s = flip ? fabsf(x) : y, t = flip ? y : fabsf(x);
    float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
    float rcp = 1.0f / (t * scale);
    float sot = (s * scale) * rcp;
    float tn = fabsf(x) == fabsf(y) ? 1.0f : fabsf(sot);
    float arc = (flip ? 1.0f : 0.0f) * 1.57079632679489661923f + m_atan_pos(tn) * m_sgn(tn);
    return fminf(y, rcp) < 0.0f ? -arc : arc;
}

/* pow(x, y) as gallivm evaluates it: exp2(y * log2(x)) with its own minimax polynomials (lp_build_exp2: degree 5 on
 * the fraction; lp_build_log2_approx: degree 4 in z = y*y, y = (m-1)/(m+1); both evaluated as even/odd split Horner
 * with fused multiply-adds, lp_build_polynomial).  Pinned against llvmpipe with a probe shader: exp2, log2 and pow
 * bit-equal on 16384/16384 random arguments.  libm powf is 2e-6 away, which advect's `1 - pow(stress, -1/12)`
 * amplifies by three orders. */
static inline float lp_exp2(float x)
{
    x = fminf(x, 128.0f);
    x = fmaxf(x, -126.99999f);
    float ip = floorf(x), fp = x - ip;
    int32_t e = ((int32_t)ip + 127) << 23;
    float scale;
    memcpy(&scale, &e, 4);
    float x2 = fp * fp;
    float odd = 0.00187757667519147912699f, even = 0.00898934009049466391101f;
    odd = fmaf(x2, odd, 0.0558263180532956664775f); even = fmaf(x2, even, 0.240153617044375388211f);
    odd = fmaf(x2, odd, 0.693153073200168932794f); even = fmaf(x2, even, 1.000000000000000000000f);
    return scale * fmaf(odd, fp, even);
}
static inline float lp_log2(float x)
{
    int32_t i;
    memcpy(&i, &x, 4);
    float logexp = (float)(((i >> 23) & 255) - 127);
    int32_t mi = (i & 0x007fffff) | 0x3f800000;
    float mant;
    memcpy(&mant, &mi, 4);
    float y = (mant - 1.0f) / (mant + 1.0f);
    float z = y * y, z2 = z * z;
    float even = 0.406718052498846252698f, odd = 0.403343858251329912514f;
    even = fmaf(z2, even, 0.577440339438736392009f); odd = fmaf(z2, odd, 0.961791550404184197881f);
    even = fmaf(z2, even, 2.88539009343309178325f);
    return fmaf(y, fmaf(odd, z, even), logexp);
}
static inline float gpow(float x, float y) { return g_glsl_math ? lp_exp2(lp_log2(x) * y) : powf(x, y); }

/* ---------- GLSL built-ins ---------- */
static inline float glsl_mod(float x, float y) { return x - y * floorf(x / y); }
static inline float glsl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
typedef struct { float x, y, z; } vec3;
static inline vec3 v3(float x, float y, float z) { vec3 r = { x, y, z }; return r; }
static inline vec3 v3_cross(vec3 a, vec3 b)
{ return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
/* Mesa lowers dot() from the last component down (pinned with probe shaders, noise_oracle.c) */
static inline float v3_dot(vec3 a, vec3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline vec3 v3_scale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v3_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3_normalize(vec3 a)
{ float r = 1.0f / sqrtf(v3_dot(a, a)); return v3(a.x * r, a.y * r, a.z * r); }  /* x * inversesqrt(dot) */

/* ---------- texture unit (src/Texture.cpp:29-33: CLAMP_TO_EDGE, NEAREST or LINEAR) ---------- */
static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

static inline float fetch_nearest(const float *t, int W, int H, int nc, int c, float u, float v)
{
    int i = clampi((int)floorf(u * (float)W), 0, W - 1);
    int j = clampi((int)floorf(v * (float)H), 0, H - 1);
    return t[((size_t)j * W + i) * nc + c];
}

static inline float fetch_linear(const float *t, int W, int H, int nc, int c, float u, float v)
{
    float fu = u * (float)W - 0.5f, fv = v * (float)H - 0.5f;
    float flu = floorf(fu), flv = floorf(fv);
    float wx = fu - flu, wy = fv - flv;
    int i0 = clampi((int)flu, 0, W - 1), i1 = clampi((int)flu + 1, 0, W - 1);
    int j0 = clampi((int)flv, 0, H - 1), j1 = clampi((int)flv + 1, 0, H - 1);
    float t00 = t[((size_t)j0 * W + i0) * nc + c], t10 = t[((size_t)j0 * W + i1) * nc + c];
    float t01 = t[((size_t)j1 * W + i0) * nc + c], t11 = t[((size_t)j1 * W + i1) * nc + c];
    /* llvmpipe's lerp is a fused multiply-add: lp_build_lerp -> a + w*(b-a) through lp_build_mad (advect goldens:
     * 699 -> 93 differing cells of 8192 with it) */
    float a = fmaf(wx, t10 - t00, t00);
    float b = fmaf(wx, t11 - t01, t01);
    return fmaf(wy, b - a, a);
}

/* tap of a GL_LINEAR texture at a texel-centre-ish coordinate (see orc_set_linear_taps) */
static inline float tap_linear(const float *t, int W, int H, int nc, int c, float u, float v)
{
    return g_bilinear ? fetch_linear(t, W, H, nc, c, u, v) : fetch_nearest(t, W, H, nc, c, u, v);
}

/* ---------- cornerCoords library (src/Shader.h:43-99) ---------- */
void orc_tex_to_spheric(const orc_grid *g, float sx, float sy, float *lon, float *lat)
{   /* :49-53 */
    *lon = (sx * (g->coords[3] - g->coords[2]) + g->coords[2]);
    *lat = (sy * (g->coords[1] - g->coords[0]) + g->coords[0]);
}

static inline void spheric_to_tex(const orc_grid *g, float lon, float lat, float *sx, float *sy)
{   /* :55-59 */
    *sx = (lon - g->coords[2]) / (g->coords[3] - g->coords[2]);
    *sy = (lat - g->coords[0]) / (g->coords[1] - g->coords[0]);
}

static inline vec3 spheric_to_cartesian(float lon, float lat)
{   /* :61-63 */
    return v3(gcos(lat) * gcos(lon), gcos(lat) * gsin(lon), gsin(lat));
}

/* pixelsize(st) :71-74 -- `vec2(dlon*cos(lat), dlat)*circumference/(2*M_PI) / textureSize(img,0)` */
void orc_pixelsize(const orc_grid *g, float sty, float *dx, float *dy)
{
    float lon, lat;
    orc_tex_to_spheric(g, 0.0f, sty, &lon, &lat);
    float two_pi = 2.0f * MPI;
    *dx = ((g->coords[3] - g->coords[2]) * gcos(lat)) * g->circumference / two_pi / (float)g->W;
    *dy = (g->coords[1] - g->coords[0]) * g->circumference / two_pi / (float)g->H;
}

/* mod(lon + 2*M_PI, 2*M_PI) - M_PI as Mesa's algebraic pass leaves it: the constants of
 * ((lon + 2pi) - 2pi*floor((lon + 2pi)/2pi)) - pi are combined into (lon + pi) - 2pi*floor((lon + 2pi)/2pi)
 * (pinned with a probe shader on 4000 pole-crossing taps with fractional offsets: 100 % bit-equal, the literal
 * form 93 %; integer offsets land on the same texels either way) */
static inline float pole_lon(float lon)
{
    const float two_pi = 2.0f * MPI;
    return (lon + MPI) - two_pi * floorf((lon + two_pi) / two_pi);
}

/* offset(p, dp, resolution) :82-98 (offset_no_globe_wrap is never set by any filter: false) */
void orc_offset(const orc_grid *g, float px, float py, float dpx, float dpy, float resx, float resy,
                float *ox, float *oy)
{
    const float *c = g->coords;
    px = px + dpx / resx;
    py = py + dpy / resy;
    int lonwrap = (c[2] < -MPI + 1e-4f) && (c[3] > MPI - 1e-3f);
    if (lonwrap) px = glsl_mod(px + 1.0f, 1.0f);
    if (c[0] < -MPI / 2.0f + 1e-4f && py < 0.0f) {
        py = -py;
        px = pole_lon(px * (c[3] - c[2]) + c[2]);
        px = (px - c[2]) / (c[3] - c[2]);
    }
    if (c[1] > MPI / 2.0f - 1e-4f && py > 1.0f) {
        py = 2.0f - py;
        px = pole_lon(px * (c[3] - c[2]) + c[2]);
        px = (px - c[2]) / (c[3] - c[2]);
    }
    if (lonwrap) px = glsl_mod(px + 1.0f, 1.0f);
    *ox = px;
    *oy = py;
}

static inline float stx(const orc_grid *g, int x) { return ((float)x + 0.5f) / (float)g->W; }
static inline float sty(const orc_grid *g, int y) { return ((float)y + 0.5f) / (float)g->H; }

/* `st` as llvmpipe's rasteriser interpolates it over the reference's full-screen quad (tools/st_probe.py: the model
 * matches a shader that writes `st` at 100 % of the pixels of every size tried).  Two triangles split along the
 * anti-diagonal, plane equations in fp32, each anchored at its own vertex; exactly (x+0.5)/W on power-of-two grids, up
 * to an ulp of 1.0 off elsewhere.  Used only where a test wants the reference's pixels on a non-power-of-two grid
 * (orc_set_raster_st); the product takes st = (x+0.5)/W, which is what the maths says and what the grids of
 * BASELINE.json give under any rasteriser. */
static int g_raster_st = 0;
void orc_set_raster_st(int on) { g_raster_st = on; }
int orc_raster_st(void) { return g_raster_st; }
void orc_st_raster(const orc_grid *g, int x, int y, float *sx, float *sy);
static inline void st_raster(const orc_grid *g, int x, int y, float *sx, float *sy) { orc_st_raster(g, x, y, sx, sy); }
void orc_st_raster_wh(int W, int H, int x, int y, float *sx, float *sy)
{
    const float Wf = (float)W, Hf = (float)H;
    const float ooa = 1.0f / (Wf * Hf);
    const float d = Hf * ooa, e = Wf * ooa;
    const int lower = ((double)x + 0.5) / W + ((double)y + 0.5) / H < 1.0;
    const float a0x = lower ? 0.0f - d * (0.0f - 0.5f) : 1.0f - d * (Wf - 0.5f);
    const float a0y = lower ? 0.0f - e * (0.0f - 0.5f) : 1.0f - e * (Hf - 0.5f);
    *sx = fmaf(d, (float)x, a0x);
    *sy = fmaf(e, (float)y, a0y);
}
void orc_st_raster(const orc_grid *g, int x, int y, float *sx, float *sy) { orc_st_raster_wh(g->W, g->H, x, y, sx, sy); }

/* ====================================================================== */
/* Ocean currents                                                          */
/* ====================================================================== */

/* One Jacobi sweep, literal: OceanCurrents.cpp:450-481.
 * pressure/divw are GL_LINEAR (:44,54), img is the Layer texture, GL_NEAREST (src/menus/LayerWindow.cpp:78-84). */
static void jacobi_sweep_literal(const orc_grid *g, const float *img, const float *divw, const float *p, float *out)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;   /* resolution = textureSize(pressure,0) */
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        float dx, dy;
        orc_pixelsize(g, sy, &dx, &dy);
        float wx = dx / 420.0f, wy = dy / 420.0f;
        float pw2x = wx * wx, pw2y = wy * wy;                 /* pow(v, vec2(2,2)) */
        float Beta = 2.0f * (1.0f / pw2x + 1.0f / pw2y);      /* :479 */
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) {                                /* per fragment, as the shader has it */
                st_raster(g, x, y, &sx, &sy);
                orc_pixelsize(g, sy, &dx, &dy);
                wx = dx / 420.0f; wy = dy / 420.0f;
                pw2x = wx * wx; pw2y = wy * wy;
                Beta = 2.0f * (1.0f / pw2x + 1.0f / pw2y);
            }
            float nx, ny, sx2, sy2, ex, ey, wx2, wy2;
            orc_offset(g, sx, sy, 0, 1, rx, ry, &nx, &ny);
            orc_offset(g, sx, sy, 0, -1, rx, ry, &sx2, &sy2);
            orc_offset(g, sx, sy, 1, 0, rx, ry, &ex, &ey);
            orc_offset(g, sx, sy, -1, 0, rx, ry, &wx2, &wy2);
            float pN = tap_linear(p, W, H, 1, 0, nx, ny);
            float pS = tap_linear(p, W, H, 1, 0, sx2, sy2);
            float pE = tap_linear(p, W, H, 1, 0, ex, ey);
            float pW = tap_linear(p, W, H, 1, 0, wx2, wy2);
            float pC = tap_linear(p, W, H, 1, 0, sx, sy);
            float oN = fetch_nearest(img, W, H, 1, 0, nx, ny);
            float oS = fetch_nearest(img, W, H, 1, 0, sx2, sy2);
            float oE = fetch_nearest(img, W, H, 1, 0, ex, ey);
            float oW = fetch_nearest(img, W, H, 1, 0, wx2, wy2);
            float oC = fetch_nearest(img, W, H, 1, 0, sx, sy);
            if (oN > 0) pN = pC;
            if (oS > 0) pS = pC;
            if (oE > 0) pE = pC;
            if (oW > 0) pW = pC;
            float bC = tap_linear(divw, W, H, 1, 0, sx, sy);
            float fc = ((pW + pE) / pw2x + (pS + pN) / pw2y - bC) / Beta;   /* :480 */
            if (oC > 0) fc = 0;
            out[(size_t)y * W + x] = fc;
        }
    }
}

/* OceanCurrents::jacobi host loop :500-508 (the caller zeroes p like :495-497 when it wants a full solve) */
void orc_ocean_jacobi(const orc_grid *g, const float *img, const float *divw, float *p, float *scratch, int iters)
{
    float *a = p, *b = scratch;
    for (int i = 0; i < iters; i++) {
        jacobi_sweep_literal(g, img, divw, a, b);
        float *t = a; a = b; b = t;                              /* pressure->swap(scratch) */
    }
    if (a != p) memcpy(p, a, sizeof(float) * (size_t)g->W * g->H);
}

/* ---- row tables + neighbour indices (the HIP kernels' contract) ----
 * Division by a row constant c is evaluated as  q = x*r ; q' = fma(fma(-c, q, x), r, q)  with r = RN(1/c)
 * (Markstein's correction step).  With a correctly rounded reciprocal this IS the IEEE quotient RN(x/c) whenever
 * the remainder does not underflow (|x| >= 2^-100 for the row constants that occur; 9e8 random and adversarial
 * operand pairs, all 2^23 divisor significands, zero mismatches); for smaller |x| it can differ from RN(x/c) by
 * one ulp, i.e. by < 1e-37.  The literal form above keeps true divisions; tests compare the two. */
static inline float div_rc(float x, float c, float r)
{
    float q = x * r;
    return fmaf(fmaf(-c, q, x), r, q);
}
void orc_jacobi_row_tables(const orc_grid *g, float *cx, float *cy, float *beta)
{
    for (int y = 0; y < g->H; y++) {
        float dx, dy;
        orc_pixelsize(g, sty(g, y), &dx, &dy);
        float wx = dx / 420.0f, wy = dy / 420.0f;
        cx[y] = wx * wx;
        cy[y] = wy * wy;
        beta[y] = 2.0f * (1.0f / cx[y] + 1.0f / cy[y]);
    }
}

void orc_pole_columns(const orc_grid *g, int *polecol)
{
    /* literal evaluation of the pole branch of offset() for a tap that leaves the grid through a pole;
     * south and north branches are the same expression in p.x (src/Shader.h:88-89 == :93-94). */
    const int W = g->W;
    for (int x = 0; x < W; x++) {
        float ox, oy;
        if (g->coords[0] < -MPI / 2.0f + 1e-4f)
            orc_offset(g, stx(g, x), sty(g, 0), 0, -1, (float)W, (float)g->H, &ox, &oy);
        else
            orc_offset(g, stx(g, x), sty(g, g->H - 1), 0, 1, (float)W, (float)g->H, &ox, &oy);
        polecol[x] = clampi((int)floorf(ox * (float)W), 0, W - 1);
    }
}

/* neighbour index rules equivalent to offset()+NEAREST for integer steps */
typedef struct { int lonwrap, spole, npole; } orc_topo;
static orc_topo topo_of(const orc_grid *g)
{
    orc_topo t;
    t.lonwrap = (g->coords[2] < -MPI + 1e-4f) && (g->coords[3] > MPI - 1e-3f);
    t.spole = g->coords[0] < -MPI / 2.0f + 1e-4f;
    t.npole = g->coords[1] > MPI / 2.0f - 1e-4f;
    return t;
}

void orc_ocean_jacobi_rows(const orc_grid *g, const float *img, const float *divw, float *p, float *scratch,
                           int iters, int nth)
{
    const int W = g->W, H = g->H;
    float *cx = malloc(sizeof(float) * H), *cy = malloc(sizeof(float) * H), *be = malloc(sizeof(float) * H);
    int *pc = malloc(sizeof(int) * W);
    orc_jacobi_row_tables(g, cx, cy, be);
    orc_pole_columns(g, pc);
    orc_topo t = topo_of(g);
    nth = nthreads_rows(H, nth);
    float *a = p, *b = scratch;
    for (int it = 0; it < iters; it++) {
#pragma omp parallel for schedule(static) num_threads(nth)
        for (int y = 0; y < H; y++) {
            const float pw2x = cx[y], pw2y = cy[y], Beta = be[y];
            const float rcx = 1.0f / pw2x, rcy = 1.0f / pw2y, rbe = 1.0f / Beta;   /* correctly rounded reciprocals */
            const float *rowC = a + (size_t)y * W, *imgC = img + (size_t)y * W;
            /* north / south rows; pole rows reflect into themselves at column polecol[x] */
            int yn = y + 1, ys = y - 1, nflip = 0, sflip = 0;
            if (yn >= H) { yn = H - 1; nflip = t.npole; }
            if (ys < 0) { ys = 0; sflip = t.spole; }
            const float *rowN = a + (size_t)yn * W, *imgN = img + (size_t)yn * W;
            const float *rowS = a + (size_t)ys * W, *imgS = img + (size_t)ys * W;
            for (int x = 0; x < W; x++) {
                int xe = x + 1, xw = x - 1;
                if (t.lonwrap) { if (xe >= W) xe = 0; if (xw < 0) xw = W - 1; }
                else { if (xe >= W) xe = W - 1; if (xw < 0) xw = 0; }
                int xn = nflip ? pc[x] : x, xs = sflip ? pc[x] : x;
                float pC = rowC[x];
                float pN = imgN[xn] > 0 ? pC : rowN[xn];
                float pS = imgS[xs] > 0 ? pC : rowS[xs];
                float pE = imgC[xe] > 0 ? pC : rowC[xe];
                float pW = imgC[xw] > 0 ? pC : rowC[xw];
                float fc = div_rc(div_rc(pW + pE, pw2x, rcx) + div_rc(pS + pN, pw2y, rcy) - divw[(size_t)y * W + x], Beta, rbe);
                if (imgC[x] > 0) fc = 0;
                b[(size_t)y * W + x] = fc;
            }
        }
        float *s = a; a = b; b = s;
    }
    if (a != p) memcpy(p, a, sizeof(float) * (size_t)W * H);
    free(cx); free(cy); free(be); free(pc);
}

/* ---- tolerance renderings of the sweep (north_star allows 1e-5 relative) ---------------------------
 * fc = ((pW + pE)/pw2.x + (pS + pN)/pw2.y - bC)/Beta (OceanCurrents.cpp:480) is linear in its operands with three row
 * constants: fc = a (pW + pE) + b (pS + pN) - c bC,  a = 1/(pw2.x Beta), b = 1/(pw2.y Beta), c = 1/Beta -- taken from the
 * reference's OWN fp32 row constants (its Beta is RN(2 (RN(1/pw2.x) + RN(1/pw2.y))), so 2a + 2b is 1 + delta, |delta| <~ 1e-7:
 * the gain of a constant field per sweep, which 5000 sweeps raise to the 5000th power) and evaluated in fp64 before rounding.
 *   form 1: a, b, c rounded to fp32; t = RN(c bC); fc = fma(a, we, fma(b, sn, -t))                       (2 adds, 1 mul, 2 fma)
 *   form 2: a = ah + al, b = bh + bl with ah + bh = 0.5 EXACTLY (the larger of the two rounded, the other 0.5 minus it:
 *           exact by Sterbenz) and the residues al, bl rounded: fc = fma(ah, we, fma(bh, sn, fma(bl, sn, fma(al, we, -t))))
 *           -- no systematic error in the gain of smooth fields                                            (2 adds, 1 mul, 4 fma)
 *   form 3: as 2 with the residues folded on the N/S sum: e = RN(al + bl): fma(ah, we, fma(bh, sn, fma(e, sn, -t)))
 *           (exact for a constant field; the E/W and N/S sums of a smooth field differ by its curvature)   (2 adds, 1 mul, 3 fma)
 *   form 4: the reference's three divisions as plain multiplications by rounded reciprocals (DESIGN.md 4's rejected form)
 * Land handling as in the exact form (a land neighbour reads the centre, a land centre is 0).
 * Test infrastructure: evaluated by tools/jacobi_tolerance_eval.py against the llvmpipe golden and the exact form. */
void orc_jacobi_fused_tables(const orc_grid *g, int form, float *tab /* [H][8]: ah, bh, al, bl, c, e, -, - */)
{
    const int H = g->H;
    float *cx = malloc(sizeof(float) * H), *cy = malloc(sizeof(float) * H), *be = malloc(sizeof(float) * H);
    orc_jacobi_row_tables(g, cx, cy, be);
    for (int y = 0; y < H; y++) {
        const double a = 1.0 / ((double)cx[y] * (double)be[y]), b = 1.0 / ((double)cy[y] * (double)be[y]);
        float *t = tab + (size_t)y * 8;
        memset(t, 0, 8 * sizeof(float));
        t[4] = (float)(1.0 / (double)be[y]);
        if (form == 1) { t[0] = (float)a; t[1] = (float)b; continue; }
        float ah, bh;
        if (a >= b) { ah = (float)a; bh = 0.5f - ah; } else { bh = (float)b; ah = 0.5f - bh; }
        t[0] = ah; t[1] = bh;
        t[2] = (float)(a - (double)ah); t[3] = (float)(b - (double)bh);
        t[5] = (float)((a - (double)ah) + (b - (double)bh));
    }
    free(cx); free(cy); free(be);
}

void orc_ocean_jacobi_rows_form(const orc_grid *g, const float *img, const float *divw, float *p, float *scratch,
                                int iters, int nth, int form)
{
    if (form == 0) { orc_ocean_jacobi_rows(g, img, divw, p, scratch, iters, nth); return; }
    const int W = g->W, H = g->H;
    float *cx = malloc(sizeof(float) * H), *cy = malloc(sizeof(float) * H), *be = malloc(sizeof(float) * H);
    float *tab = malloc(sizeof(float) * 8 * H);
    int *pc = malloc(sizeof(int) * W);
    orc_jacobi_row_tables(g, cx, cy, be);
    orc_jacobi_fused_tables(g, form == 4 ? 1 : form, tab);
    orc_pole_columns(g, pc);
    orc_topo t = topo_of(g);
    nth = nthreads_rows(H, nth);
    float *a = p, *b = scratch;
    for (int it = 0; it < iters; it++) {
#pragma omp parallel for schedule(static) num_threads(nth)
        for (int y = 0; y < H; y++) {
            const float *k = tab + (size_t)y * 8;
            const float ah = k[0], bh = k[1], al = k[2], bl = k[3], c = k[4], e = k[5];
            const float rcx = 1.0f / cx[y], rcy = 1.0f / cy[y], rbe = 1.0f / be[y];
            const float *rowC = a + (size_t)y * W, *imgC = img + (size_t)y * W;
            int yn = y + 1, ys = y - 1, nflip = 0, sflip = 0;
            if (yn >= H) { yn = H - 1; nflip = t.npole; }
            if (ys < 0) { ys = 0; sflip = t.spole; }
            const float *rowN = a + (size_t)yn * W, *imgN = img + (size_t)yn * W;
            const float *rowS = a + (size_t)ys * W, *imgS = img + (size_t)ys * W;
            for (int x = 0; x < W; x++) {
                int xe = x + 1, xw = x - 1;
                if (t.lonwrap) { if (xe >= W) xe = 0; if (xw < 0) xw = W - 1; }
                else { if (xe >= W) xe = W - 1; if (xw < 0) xw = 0; }
                int xn = nflip ? pc[x] : x, xs = sflip ? pc[x] : x;
                float pC = rowC[x];
                float pN = imgN[xn] > 0 ? pC : rowN[xn];
                float pS = imgS[xs] > 0 ? pC : rowS[xs];
                float pE = imgC[xe] > 0 ? pC : rowC[xe];
                float pW = imgC[xw] > 0 ? pC : rowC[xw];
                const float we = pW + pE, sn = pS + pN, bC = divw[(size_t)y * W + x];
                float fc;
                if (form == 4) fc = ((we * rcx + sn * rcy) - bC) * rbe;
                else {
                    const float tt = c * bC;
                    if (form == 1) fc = fmaf(ah, we, fmaf(bh, sn, -tt));
                    else if (form == 2) fc = fmaf(ah, we, fmaf(bh, sn, fmaf(bl, sn, fmaf(al, we, -tt))));
                    else fc = fmaf(ah, we, fmaf(bh, sn, fmaf(e, sn, -tt)));
                }
                if (imgC[x] > 0) fc = 0;
                b[(size_t)y * W + x] = fc;
            }
        }
        float *s = a; a = b; b = s;
    }
    if (a != p) memcpy(p, a, sizeof(float) * (size_t)W * H);
    free(cx); free(cy); free(be); free(tab); free(pc);
}

/* get_velocity(st,o) of divergence/diffusion shaders (OceanCurrents.cpp:376-393, :647-661).
 * Returns v (sign-flipped across a pole) and the neighbour coordinate. */
static inline void get_velocity(const orc_grid *g, const float *v, float sx, float sy, float ox, float oy,
                                float *vx, float *vy, float *sox, float *soy)
{
    const int W = g->W, H = g->H;
    float px, py;
    orc_offset(g, sx, sy, ox, oy, (float)W, (float)H, &px, &py);
    float lon, lat, lon_o, lat_o;
    orc_tex_to_spheric(g, sx, sy, &lon, &lat);
    orc_tex_to_spheric(g, px, py, &lon_o, &lat_o);
    float a = tap_linear(v, W, H, 2, 0, px, py), b = tap_linear(v, W, H, 2, 1, px, py);
    if (fabsf(fabsf(lon - lon_o) - 3.14159f) < 0.1f) { a = -a; b = -b; }   /* :389 / :657 */
    *vx = a; *vy = b; *sox = px; *soy = py;
}

/* divergence shader :376-420 */
void orc_ocean_divergence(const orc_grid *g, const float *img, const float *v, float pressurefactor, float *out)
{
    const int W = g->W, H = g->H;
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        float dxc, dyc;
        orc_pixelsize(g, sy, &dxc, &dyc);
        float pwx = dxc / 420.0f, pwy = dyc / 420.0f;      /* :419 */
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_pixelsize(g, sy, &dxc, &dyc); pwx = dxc / 420.0f; pwy = dyc / 420.0f; }
            float vel[4][2], so[4][2];
            const float offs[4][2] = { { 0, 1 }, { 0, -1 }, { 1, 0 }, { -1, 0 } };   /* N S E W */
            float o[4];
            for (int k = 0; k < 4; k++) {
                get_velocity(g, v, sx, sy, offs[k][0], offs[k][1], &vel[k][0], &vel[k][1], &so[k][0], &so[k][1]);
                float ddx, ddy;
                orc_pixelsize(g, so[k][1], &ddx, &ddy);                 /* d_o = pixelsize(st_o) */
                float area = ddx * ddy;
                vel[k][0] = vel[k][0] * area * pressurefactor;          /* :392 */
                vel[k][1] = vel[k][1] * area * pressurefactor;
                o[k] = fetch_nearest(img, W, H, 1, 0, so[k][0], so[k][1]);
            }
            float vNy = vel[0][1], vSy = vel[1][1], vEx = vel[2][0], vWx = vel[3][0];
            if (o[0] > 0) vNy = 0;
            if (o[1] > 0) vSy = 0;
            if (o[2] > 0) vEx = 0;
            if (o[3] > 0) vWx = 0;
            out[(size_t)y * W + x] = 0.5f * ((vEx - vWx) / pwx + (vNy - vSy) / pwy);   /* :420 */
        }
    }
}

/* diffusion shader :664-693, `iters` ping-pong sweeps :701-707 */
void orc_ocean_diffusion(const orc_grid *g, const float *img, float *v, float *scratch, int iters, int scalar_rhs)
{
    const int W = g->W, H = g->H;
    float *a = v, *b = scratch;
    for (int it = 0; it < iters; it++) {
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
        for (int y = 0; y < H; y++) {
            float sy = sty(g, y);
            float dx, dy;
            orc_pixelsize(g, sy, &dx, &dy);
            /* :689  1*1/pow(pixelsize(st),vec2(2,2))*420*420 and :690 Beta.  Mesa's algebraic pass merges the two
             * constants, pulls the common factor out of pw2.x + pw2.y, and associates the final sum as (bC + a) + b
             * (pinned with a probe shader: each form 100 % bit-equal on 16384 random inputs, the literal forms 66-76 %) */
            float ix = 1.0f / (dx * dx), iy = 1.0f / (dy * dy);
            float px2 = ix * 176400.0f, py2 = iy * 176400.0f;
            float s = (ix + iy) * 352800.0f;
            float Beta = s * (1.0f + 1.0f / s);          /* :690 */
            for (int x = 0; x < W; x++) {
                float sx = stx(g, x);
                if (g_raster_st) {                       /* per fragment, as the shader has it */
                    st_raster(g, x, y, &sx, &sy);
                    orc_pixelsize(g, sy, &dx, &dy);
                    ix = 1.0f / (dx * dx); iy = 1.0f / (dy * dy);
                    px2 = ix * 176400.0f; py2 = iy * 176400.0f;
                    s = (ix + iy) * 352800.0f;
                    Beta = s * (1.0f + 1.0f / s);
                }
                const float offs[4][2] = { { 0, 1 }, { 0, -1 }, { 1, 0 }, { -1, 0 } };
                float pv[4][2], so[4][2];
                float pCx = tap_linear(a, W, H, 2, 0, sx, sy), pCy = tap_linear(a, W, H, 2, 1, sx, sy);
                for (int k = 0; k < 4; k++) {
                    get_velocity(g, a, sx, sy, offs[k][0], offs[k][1], &pv[k][0], &pv[k][1], &so[k][0], &so[k][1]);
                    if (fetch_nearest(img, W, H, 1, 0, so[k][0], so[k][1]) > 0) { pv[k][0] = pCx; pv[k][1] = pCy; }
                }
                float bCx = pCx, bCy = scalar_rhs ? pCx : pCy;     /* :687 `float bC = texture(v,st).r` */
                /* :693  ((pW + pE)*pw2.x + (pS + pN)*pw2.y - (-bC)) / Beta */
                b[((size_t)y * W + x) * 2 + 0] = ((bCx + (pv[3][0] + pv[2][0]) * px2) + (pv[1][0] + pv[0][0]) * py2) / Beta;
                b[((size_t)y * W + x) * 2 + 1] = ((bCy + (pv[3][1] + pv[2][1]) * px2) + (pv[1][1] + pv[0][1]) * py2) / Beta;
            }
        }
        float *t = a; a = b; b = t;
    }
    if (a != v) memcpy(v, a, sizeof(float) * 2 * (size_t)W * H);
}

/* v_to_cartesian / cartesian_to_v (OceanCurrents.cpp:251-268) */
static inline vec3 v_to_cartesian(float vx, float vy, float lon, float lat)
{
    vec3 c = spheric_to_cartesian(lon, lat);
    float il = 1.0f / sqrtf(c.y * c.y + c.x * c.x);      /* normalize(c.xy): dot() from the last component down */
    float inx = c.x * il, iny = c.y * il;                                   /* normalize(c.xy) */
    vec3 ycomp = v3(gsin(lat) * (-inx), gsin(lat) * (-iny), gcos(lat));
    vec3 par = v3_normalize(v3_cross(v3(0, 0, 1), c));
    return v3_add(v3_scale(par, vx), v3_scale(ycomp, vy));
}
static inline void cartesian_to_v(vec3 v, float lon, float lat, float *ox, float *oy)
{
    vec3 c = spheric_to_cartesian(lon, lat);
    float il = 1.0f / sqrtf(c.y * c.y + c.x * c.x);      /* normalize(c.xy): dot() from the last component down */
    float inx = c.x * il, iny = c.y * il;
    vec3 ycomp = v3(gsin(lat) * (-inx), gsin(lat) * (-iny), gcos(lat));
    vec3 par = v3_normalize(v3_cross(v3(0, 0, 1), c));
    *ox = v3_dot(v, par);
    *oy = v3_dot(v, ycomp);
}
/* rotation_matrix(theta,u)*v (src/Shader.h:35-40): mat3 is column-major, M*v = c0*v.x + c1*v.y + c2*v.z */
static inline vec3 rotate(float theta, vec3 u, vec3 v)
{
    float c = gcos(theta), s = gsin(theta), ic = 1.0f - c;
    vec3 c0 = v3(c + u.x * u.x * ic, u.y * u.x * ic + u.z * s, u.z * u.x * ic - u.y * s);
    vec3 c1 = v3(u.x * u.y * ic - u.z * s, c + u.y * u.y * ic, u.z * u.y * ic + u.x * s);
    vec3 c2 = v3(u.x * u.z * ic + u.y * s, u.y * u.z * ic - u.x * s, c + u.z * u.z * ic);
    return v3_add(v3_add(v3_scale(c0, v.x), v3_scale(c1, v.y)), v3_scale(c2, v.z));
}

/* advect shader :282-348 */
static int g_advect_stage = 0;      /* debugging aid for probe shaders: which intermediate to write instead of fc */
void orc_set_advect_stage(int s) { g_advect_stage = s; }
void orc_ocean_advect(const orc_grid *g, const float *img, const float *v, float *out)
{
    const int W = g->W, H = g->H;
    const float k04 = powf(0.4f, 1.0f / 24.0f);          /* pow(vec2(0.4),vec2(1.0/24)) :348 */
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) st_raster(g, x, y, &sx, &sy);
            float *o = out + ((size_t)y * W + x) * 2;
            if (fetch_nearest(img, W, H, 1, 0, sx, sy) > 0) { o[0] = 0; o[1] = 0; continue; }   /* :284-288 */
            float velx = tap_linear(v, W, H, 2, 0, sx, sy), vely = tap_linear(v, W, H, 2, 1, sx, sy);   /* :290 */
            float timestep = 24.0f;
            float distance = sqrtf(velx * velx + vely * vely) * timestep;          /* :298 */
            float arclength = 2.0f * 3.14159f / g->circumference * distance;       /* :299 */
            float lon, lat;
            orc_tex_to_spheric(g, sx, sy, &lon, &lat);
            vec3 cart = spheric_to_cartesian(lon, lat);
            vec3 v_cart = v_to_cartesian(velx, vely, lon, lat);
            vec3 rotdir = v3_normalize(v3_cross(cart, v_cart));                    /* :306 */
            cart = rotate(-arclength, rotdir, cart);                               /* :308 */
            if (g_advect_stage == 4) { o[0] = arclength; o[1] = rotdir.x; continue; }
            if (g_advect_stage == 5) { o[0] = rotdir.y; o[1] = rotdir.z; continue; }
            if (g_advect_stage == 6) { o[0] = v_cart.x; o[1] = v_cart.y; continue; }
            if (g_advect_stage == 7) { o[0] = v_cart.z; o[1] = cart.z; continue; }
            if (g_advect_stage == 8) { o[0] = cart.x; o[1] = cart.y; continue; }
            float lon_n = gatan2(cart.y, cart.x), lat_n = gasin(cart.z);           /* cartesian_to_spheric */
            if (g_advect_stage == 9) { o[0] = lon_n; o[1] = lat_n; continue; }
            float cx, cy;
            spheric_to_tex(g, lon_n, lat_n, &cx, &cy);
            if (g_advect_stage == 2) { o[0] = cx; o[1] = cy; continue; }
            float nvx = fetch_linear(v, W, H, 2, 0, cx, cy), nvy = fetch_linear(v, W, H, 2, 1, cx, cy);   /* :315 */
            if (g_advect_stage == 3) { o[0] = nvx; o[1] = nvy; continue; }
            vec3 testv = v_to_cartesian(nvx, nvy, lon_n, lat_n);
            testv = rotate(arclength, rotdir, testv);
            cartesian_to_v(testv, lon, lat, &nvx, &nvy);
            if (isnan(nvx) || isnan(nvy)) { nvx = 0; nvy = 0; }                    /* :321 */
            /* Coriolis term is multiplied by 0.0 (:327); NaN/Inf cannot arise from it once newV is finite */
            float fcx = 1.0f * nvx, fcy = 1.0f * nvy;                              /* Dissipation :332 */
            if (g_advect_stage == 1) { o[0] = fcx; o[1] = fcy; continue; }
            float phi = 2.0f * (sy - 0.5f) * 3.14159f;                             /* :341 */
            /* phi*3/2: Mesa merges the literal factors of the chain 2*(.)*3.14159*3/2 into one constant, and evaluates
             * 0.0001*pow(a,2) as (0.0001*a)*a (probe shaders, tools/advect_probe.py: 100 % of cells either way) */
            float phi32 = (sy - 0.5f) * ((2.0f * 3.14159f) * (3.0f * 0.5f));
            float windx = 10.0f * -gcos(phi32), windy = 10.0f * gsin(phi32);
            if (fabsf(phi32) > 3.14159f) windx = -windx;
            if (fabsf(phi) > 3.14159f * 1.0f / 3.0f && fabsf(phi) < 3.1459f * 2.0f / 3.0f) windy = -windy;   /* sic 3.1459 */
            if (phi < 0) windy = -windy;
            float ax = fabsf(windx - fcx), ay = fabsf(windy - fcy);
            float stressx = 1.0f + (0.0001f * ax) * ax, stressy = 1.0f + (0.0001f * ay) * ay;   /* :347 */
            float e = -1.0f / 24.0f * 2.0f;                                        /* -vec2(1.0)/24.0*2 */
            if (g_advect_stage == 10) { o[0] = stressx; o[1] = stressy; continue; }
            if (g_advect_stage == 11) { o[0] = gpow(stressx, e); o[1] = gpow(stressy, e); continue; }
            if (g_advect_stage == 12) { o[0] = windx; o[1] = windy; continue; }
            if (g_advect_stage == 13) { o[0] = windx * (1.0f - gpow(stressx, e)); o[1] = windy * (1.0f - gpow(stressy, e)); continue; }
            if (g_advect_stage == 14) { o[0] = fcx * (1.0f - k04); o[1] = fcy * (1.0f - k04); continue; }
            o[0] = fcx + windx * (1.0f - gpow(stressx, e)) - fcx * (1.0f - k04);    /* :348 */
            o[1] = fcy + windy * (1.0f - gpow(stressy, e)) - fcy * (1.0f - k04);
        }
    }
}

/* subDiv shader :532-606 */
void orc_ocean_subdiv(const orc_grid *g, const float *img, const float *v, const float *pressure,
                      float pressurefactor, float *out)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    static const float offsets[8][2] = { { 1, 0 }, { 1, 1 }, { 0, 1 }, { -1, 1 }, { -1, 0 }, { -1, -1 }, { 0, -1 }, { 1, -1 } };  /* :560 */
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        float dx, dy;
        orc_pixelsize(g, sy, &dx, &dy);
        float pwx = dx / 420.0f, pwy = dy / 420.0f;      /* :574 */
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_pixelsize(g, sy, &dx, &dy); pwx = dx / 420.0f; pwy = dy / 420.0f; }
            float *o = out + ((size_t)y * W + x) * 2;
            if (fetch_nearest(img, W, H, 1, 0, sx, sy) > 0) { o[0] = 0; o[1] = 0; continue; }   /* :536-540 */
            float c[4][2];
            const float offs[4][2] = { { 0, 1 }, { 0, -1 }, { 1, 0 }, { -1, 0 } };
            float pn[4], on[4];
            float pC = tap_linear(pressure, W, H, 1, 0, sx, sy);
            for (int k = 0; k < 4; k++) {
                orc_offset(g, sx, sy, offs[k][0], offs[k][1], rx, ry, &c[k][0], &c[k][1]);
                pn[k] = tap_linear(pressure, W, H, 1, 0, c[k][0], c[k][1]);
                on[k] = fetch_nearest(img, W, H, 1, 0, c[k][0], c[k][1]);
                if (on[k] > 0) pn[k] = pC;                                          /* :566-569 */
            }
            float oldx = tap_linear(v, W, H, 2, 0, sx, sy), oldy = tap_linear(v, W, H, 2, 1, sx, sy);
            /* :575  oldV - 0.5*vec2(pE-pW,pN-pS)/pixelwidth / pixelsize.x / pixelsize.y / pressurefactor */
            float fx = oldx - 0.5f * (pn[2] - pn[3]) / pwx / dx / dy / pressurefactor;
            float fy = oldy - 0.5f * (pn[0] - pn[1]) / pwy / dx / dy / pressurefactor;
            float o_array[8];
            for (int i = 0; i < 8; i++) {
                float qx, qy;
                orc_offset(g, sx, sy, offsets[i][0], offsets[i][1], rx, ry, &qx, &qy);
                o_array[i] = fetch_nearest(img, W, H, 1, 0, qx, qy);
            }
            float ang = (gatan2(fy, fx) / MPI + 1.0f) / 2.0f * 8.0f + 4.0f;
            float lower = glsl_mod(floorf(ang), 8.0f);
            float upper = glsl_mod(ceilf(ang), 8.0f);
            float theta = glsl_mod(ang, 8.0f);
            int isBorder = o_array[clampi((int)lower, 0, 7)] > 0 || o_array[clampi((int)upper, 0, 7)] > 0;
            float nx = fx, ny = fy;
            float difference = 2.0f * MPI;
            for (int i = 0; i < 8; i++) {
                float ox = offsets[i][0], oy = offsets[i][1];
                /* compass offsets are constants: the 8-iteration loop unrolls and Mesa folds this atan on the host */
                float thetai = glsl_mod((atan2f(oy, ox) / MPI + 1.0f) / 2.0f * 8.0f + 4.0f, 8.0f);
                float d = fabsf(thetai - theta);
                float angleToVel = fminf((2.0f * MPI) - d, d);
                if (angleToVel < difference && o_array[clampi((int)roundf(thetai), 0, 7)] <= 0) {
                    difference = angleToVel;
                    float lo = sqrtf(ox * ox + oy * oy), ln = sqrtf(nx * nx + ny * ny);
                    nx = ox / lo * ln;
                    ny = oy / lo * ln;
                }
            }
            if (isBorder) { fx = nx; fy = ny; }
            o[0] = fx; o[1] = fy;
        }
    }
}

/* ====================================================================== */
/* Gaussian blur (src/filter/BlurMenu.cpp)                                 */
/* ====================================================================== */
int orc_blur_rlist(float radius, float *rlist, int cap)
{
    /* :27 radius = radius/2;  :79-94 schedule, fp32 with the reference's implicit double sqrt */
    radius = radius / 2;
    float R = radius * radius / 2;
    int n = 0;
    float i = 1.0f, incrementer = 0.5f;
    if (R < 3) {
        float k = (float)(1 / sqrt((double)(55 / R)));
        incrementer = k;
        i = k;
    }
    while (R >= i * i) {
        R -= i * i;
        if (n < cap) rlist[n] = i;
        n++;
        i += incrementer;
    }
    if (R > 0.0f) { if (n < cap) rlist[n] = (float)sqrt((double)R); n++; }
    /* std::sort ascending :94 */
    int m = n < cap ? n : cap;
    for (int a = 1; a < m; a++) { float k = rlist[a]; int b = a - 1; while (b >= 0 && rlist[b] > k) { rlist[b + 1] = rlist[b]; b--; } rlist[b + 1] = k; }
    return n;
}

/* what a "per-row weights" rendering of the blur would differ by.  The literal shader reaches a tap through
 * offset() in fp32 texture coordinates: st.x + off.x / W carries ~W * 2^-24 texels of rounding noise, so a tap meant to sit ON a
 * texel centre across the pass's axis blends in ~1e-3 (W = 16384) of the neighbouring column, and the weights along the axis move
 * by as much from cell to cell.  orc_set_blur_clean(1) evaluates every tap at the position the arithmetic MEANS -- the fractional
 * part of the offset itself (one pair of weights per row and tap), exactly on the centre across the axis -- by moving the literal
 * coordinate to the nearest such position; tools/blur_tolerance_eval.py measures the difference. */
static int g_blur_clean = 0;
void orc_set_blur_clean(int on) { g_blur_clean = on; }
static inline float blur_clean_coord(float u, int n, float off_texels)
{
    const double fu = (double)u * n - 0.5, f = fabs((double)off_texels) - floor(fabs((double)off_texels));
    const double base = floor(fu);
    double best = fu, bd = 1e9;
    for (int k = -1; k <= 1; k++)
        for (int s = 0; s < 2; s++) {
            const double c = base + k + (s ? 1.0 - f : f);
            if (fabs(c - fu) < bd) { bd = fabs(c - fu); best = c; }
        }
    return (float)((best + 0.5) / n);
}
static inline float blur_tap(const orc_grid *g, const float *src, float sx, float sy, float ox, float oy)
{
    const int W = g->W, H = g->H;
    float px, py;
    orc_offset(g, sx, sy, ox, oy, (float)W, (float)H, &px, &py);
    if (g_blur_clean) { px = blur_clean_coord(px, W, ox); py = blur_clean_coord(py, H, oy); }
    return fetch_linear(src, W, H, 1, 0, px, py);
}

/* blur13 :41-62; image is GL_LINEAR (:29-30): genuine bilinear taps at fractional offsets */
void orc_blur13_pass(const orc_grid *g, const float *src, float dirx, float diry, float *dst)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    const float k1 = 1.411764705882353f, k2 = 3.2941176470588234f, k3 = 5.176470588235294f;
    const float w0 = 0.1964825501511404f, w1 = 0.2969069646728344f, w2 = 0.09447039785044732f, w3 = 0.010381362401148057f;
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        float lon, lat;
        orc_tex_to_spheric(g, 0.0f, sy, &lon, &lat);
        float phifactor = gcos(fabsf(lat));                                   /* :49 */
        float o1x = k1 * dirx / phifactor, o2x = k2 * dirx / phifactor, o3x = k3 * dirx / phifactor;
        float o1y = k1 * diry, o2y = k2 * diry, o3y = k3 * diry;
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) {                           /* per fragment, as the shader has it */
                st_raster(g, x, y, &sx, &sy);
                orc_tex_to_spheric(g, 0.0f, sy, &lon, &lat);
                phifactor = gcos(fabsf(lat));
                o1x = k1 * dirx / phifactor; o2x = k2 * dirx / phifactor; o3x = k3 * dirx / phifactor;
            }
            float color = 0.0f;
            (void)rx; (void)ry;
            color += fetch_linear(src, W, H, 1, 0, sx, sy) * w0;
            color += blur_tap(g, src, sx, sy, o1x, o1y) * w1;
            color += blur_tap(g, src, sx, sy, -o1x, -o1y) * w1;
            color += blur_tap(g, src, sx, sy, o2x, o2y) * w2;
            color += blur_tap(g, src, sx, sy, -o2x, -o2y) * w2;
            color += blur_tap(g, src, sx, sy, o3x, o3y) * w3;
            color += blur_tap(g, src, sx, sy, -o3x, -o3y) * w3;
            dst[(size_t)y * W + x] = color;
        }
    }
}

void orc_blur(const orc_grid *g, float *field, float *scratch, float radius)
{
    float rl[4096];
    int n = orc_blur_rlist(radius, rl, 4096);
    /* Blur::step :97-117: vertical tex1->tex2 with direction (0,r), horizontal tex2->tex1 with (r,0);
     * `i>rlist.size()-1` is an unsigned compare: an empty rlist never happens for radius>0 */
    int i = 0;
    do {
        orc_blur13_pass(g, field, 0.0f, rl[i], scratch);
        orc_blur13_pass(g, scratch, rl[i], 0.0f, field);
        i++;
    } while (!(i > n - 1));
}

/* ====================================================================== */
/* Climate (src/filter/Temperature.cpp:56-134)                             */
/* ====================================================================== */
/* Every literal of Temperature.cpp as a field (include/demiurge_hip.h dmg_climate_params has the same list, with the
 * lines); orc_climate_params_default() = the reference.  The arithmetic keeps the shader's form whatever the values, so a
 * parity test can move them (a stable year on a fine grid needs other values than the reference's). */
void orc_climate_params_default(orc_climate_params *p)
{
    p->steps_per_year = 15000; p->year_seconds = 3.154f * 1e7f; p->diffusivity = 0.55f * 1e6f; p->albedo = 0.30f;
    p->olr_coeff = 210.0f; p->olr_ref_kelvin = 273.4f; p->emissivity = 0.93f; p->celsius_offset = 273.15f;
    p->heat_capacity_land = 1e7f + 1e7f * 0.5f; p->heat_capacity_ocean = 1e7f + 4.0f * 1.5f * 1e7f;
    p->eccentricity = 0.017f; p->obliquity_deg = 23.44f; p->perihelion_deg = 77.05f; p->solar_constant = 1365.0f;
    p->initial_temperature = 0.0f;
    p->quirks = 1;
}

/* returns ASR = (1-0.30)*QDay(phi,M) in the association Mesa gives it; parts[5] is QDay itself */
static float qday_parts_p(const orc_climate_params *cp, float phi, float M, float *parts)
{
    const float eccentricity = cp->eccentricity;
    const float gamma = cp->obliquity_deg / 180.0f * MPI;
    const float omega = 0.0f, omega2 = cp->perihelion_deg / 180.0f * MPI;
    const float S0 = cp->solar_constant;
    /* A(M) :73-75 */
    float e3 = eccentricity * eccentricity * eccentricity;   /* pow(e,3) */
    float A = M + (2.0f * eccentricity - e3 / 4.0f * gsin(M) + 5.0f / 4.0f * (eccentricity * eccentricity) * gsin(2.0f * M)
                   + 13.0f / 12.0f * e3 * gsin(3.0f * M));
    float Ls = A - omega2;
    float delt = gasin(sinf(gamma) * gsin(Ls));          /* sin(gamma) is a constant expression: folded on the host */
    /* h0 :85-89 */
    float h = glsl_sign(phi) == glsl_sign(delt) ? MPI : 0.0f;
    if (fabsf(phi) <= MPI / 2.0f - fabsf(delt)) {
        float arg = -gtan(phi) * gtan(delt);
        if (!(cp->quirks & 1)) arg = fminf(fmaxf(arg, -1.0f), 1.0f);
        h = gacos(arg);
    }
    float S = S0 * (1.0f + 2.0f * eccentricity * gcos(A - omega));
    float X = h * gsin(phi) * gsin(delt) + gcos(phi) * gcos(delt) * gsin(h);
    float Q = S / MPI * X;
    if (parts) { parts[0] = A; parts[1] = delt; parts[2] = h; parts[3] = S; parts[4] = X; parts[5] = Q; parts[6] = ((S / MPI) * (1.0f - cp->albedo)) * X; }
    /* :122 (1-alpha)*QDay: Mesa moves the constant next to the division, ((S/pi)*0.7)*X (tools/climate_probe.py) */
    return ((S / MPI) * (1.0f - cp->albedo)) * X;
}
/* debugging aid for probe shaders (tools/climate_probe.py): the intermediates of QDay for one latitude and M */
void orc_qday_parts(float phi, float M, float *parts7)
{
    orc_climate_params cp;
    orc_climate_params_default(&cp);
    qday_parts_p(&cp, phi, M, parts7);
}

void orc_temperature_p(const orc_grid *g, float *T, float *scratch, const float *terrain, const orc_climate_params *cp, int first_step, int n_steps)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    float *a = T, *b = scratch;
    const float ref_t4 = powf(cp->olr_ref_kelvin, 4.0f);       /* pow(273.4,4): a constant expression, folded by Mesa with libm */
    const float steps_f = (float)cp->steps_per_year;
    for (int i = first_step; i < first_step + n_steps; i++) {
        float M = (float)(M_PI * 2 / cp->steps_per_year * i);          /* :146 glUniform1f(id, M_PI*2/15000*i) */
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
        for (int y = 0; y < H; y++) {
            float sy = sty(g, y);
            float lon, phi;
            orc_tex_to_spheric(g, 0.0f, sy, &lon, &phi);
            float ASR = qday_parts_p(cp, phi, M, 0);        /* :102,:122 (1-0.30)*QDay(phi,M) */
            float factor = gcos(phi);                      /* get_texture_laplacian, Shader.h:301-302 */
            float dx, dy;
            orc_pixelsize(g, sy, &dx, &dy);
            float den = 4.0f * (dy * dy);                  /* 4*pow(pixelwidth.y,2) */
            for (int x = 0; x < W; x++) {
                float sx = stx(g, x);
                if (g_raster_st) {                       /* per fragment, as the shader has it */
                    st_raster(g, x, y, &sx, &sy);
                    orc_tex_to_spheric(g, 0.0f, sy, &lon, &phi);
                    ASR = qday_parts_p(cp, phi, M, 0);
                    factor = gcos(phi);
                    orc_pixelsize(g, sy, &dx, &dy);
                    den = 4.0f * (dy * dy);
                }
                float Tc = fetch_nearest(a, W, H, 1, 0, sx, sy);
                float terr = fetch_nearest(terrain, W, H, 1, 0, sx, sy);
                float s[9];
                /* Shader.h:304-312: taps offset(st,-vec2(i/factor, j)) for a..i */
                const float ox[9] = { 1, 0, -1, 1, 0, -1, 1, 0, -1 };
                const float oy[9] = { 1, 1, 1, 0, 0, 0, -1, -1, -1 };
                for (int k = 0; k < 9; k++) {
                    float px, py;
                    float dpx = ox[k] == 0 ? -0.0f : -(ox[k] / factor);
                    orc_offset(g, sx, sy, dpx, -oy[k], rx, ry, &px, &py);
                    s[k] = fetch_nearest(a, W, H, 1, 0, px, py);
                }
                float A_ = s[0], B_ = s[1], C_ = s[2], D_ = s[3], E_ = s[4], F_ = s[5], G_ = s[6], H_ = s[7], I_ = s[8];
                float delta_x = (A_ - 2 * B_ + C_ + 2 * D_ - 4 * E_ + 2 * F_ + G_ - 2 * H_ + I_) / den;
                float delta_y = (A_ + 2 * B_ + C_ - 2 * D_ - 4 * E_ - 2 * F_ + G_ + 2 * H_ + I_) / den;
                if (isnan(delta_x)) delta_x = 0;
                if (isnan(delta_y)) delta_y = 0;
                float t = Tc + cp->celsius_offset;
                float t2 = t * t;                                             /* pow(x,4) -> (x*x)*(x*x) */
                float OLR = t2 * (t2 * cp->olr_coeff) / ref_t4 * cp->emissivity;   /* :125, in Mesa's association (tools/climate_probe.py) */
                float change = ASR - OLR + cp->diffusivity * (delta_x + delta_y);   /* :128 */
                /* C = atmosphere + (terrain>0 ? atmosphere*0.5 : 4*1.5*atmosphere), atmosphere = 1e7 (:131-132) */
                float C = terr > 0 ? cp->heat_capacity_land : cp->heat_capacity_ocean;
                b[(size_t)y * W + x] = Tc + change * cp->year_seconds / steps_f / C;   /* :133, the two constant factors merged */
            }
        }
        float *t = a; a = b; b = t;
    }
    if (a != T) memcpy(T, a, sizeof(float) * (size_t)W * H);
}

void orc_temperature(const orc_grid *g, float *T, float *scratch, const float *terrain, int first_step, int n_steps)
{
    orc_climate_params cp;
    orc_climate_params_default(&cp);
    orc_temperature_p(g, T, scratch, terrain, &cp, first_step, n_steps);
}

/* ====================================================================== */
/* Erosion / uplift pass (src/filter/cpufilter.cpp:121-198)                 */
/* ====================================================================== */
void orc_erosion_pass(const orc_grid *g, const float *hgt, const float *flow, const float *updrift,
                      float factor, float slope_exponent, float *out)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    static const float offs[8][2] = { { 1, 1 }, { 0, 1 }, { -1, 1 }, { 1, 0 }, { -1, 0 }, { 1, -1 }, { 0, -1 }, { -1, -1 } };  /* :134-189 order */
    const float SLOPE = tanf(MPI / 2.0f / 3.0f);          /* :191 */
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        float psx, psy;
        orc_pixelsize(g, sy, &psx, &psy);
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_pixelsize(g, sy, &psx, &psy); }
            float h = fetch_nearest(hgt, W, H, 1, 0, sx, sy);
            if (h <= 0) { out[(size_t)y * W + x] = h; continue; }        /* :122-123 */
            float maxslope = 0;
            float dist = sqrtf(psx * psx + psy * psy);                    /* length(psize) :128 */
            for (int k = 0; k < 8; k++) {
                float px, py;
                orc_offset(g, sx, sy, offs[k][0], offs[k][1], rx, ry, &px, &py);
                float h2 = fetch_nearest(hgt, W, H, 1, 0, px, py);
                float ax = psx * offs[k][0], ay = psy * offs[k][1];
                float ndist = sqrtf(ax * ax + ay * ay);
                h2 = (h - h2) / ndist;
                if (h2 > maxslope) { maxslope = h2; dist = ndist; }
            }
            float hdiff = SLOPE * dist - maxslope * dist;                  /* :192 */
            /* :195 factor*4*flow*pow(maxslope,n)/pow(0.1,n)*0.1 */
            float fl = factor * 4.0f * fetch_nearest(flow, W, H, 1, 0, sx, sy) * gpow(maxslope, slope_exponent)
                       / gpow(0.1f, slope_exponent) * 0.1f;
            float up = fetch_nearest(updrift, W, H, 1, 0, sx, sy);
            out[(size_t)y * W + x] = h + fminf(hdiff, fmaxf(0.0f, up - fl));   /* :198 */
        }
    }
}

/* ====================================================================== */
/* Flow accumulation: the GLSL passes of FlowFilter (src/filter/FlowFilter.cpp) */
/* ====================================================================== */
static inline float glsl_fract(float x) { return x - floorf(x); }
/* hash()/noise() of the direction shader (:114-131).  noise() is only ever called at integer lattice points
 * (st*resolution*2 = (2x+1, 2y+1)), where it reduces to hash(i) exactly; the general form is kept. */
/* hash() and mix() in the forms Mesa leaves them (probe shaders: 131072/131072; the literal forms 99.4 % / 57 %):
 * p.x + p.y with p = 50*fract(..) is factored into 50*(fract_x + fract_y); mix(a,b,t) is lowered to a + t*(b - a) */
static inline float flow_hash(float px, float py)
{
    float fx = glsl_fract(px * 0.3183099f + 0.71f), fy = glsl_fract(py * 0.3183099f + 0.113f);
    px = 50.0f * fx;
    py = 50.0f * fy;
    return -1.0f + 2.0f * glsl_fract((px * py) * (50.0f * (fx + fy)));
}
static inline float flow_mix(float a, float b, float t) { return a + t * (b - a); }
static inline float flow_noise(float px, float py)
{
    float ix = floorf(px), iy = floorf(py), fx = glsl_fract(px), fy = glsl_fract(py);
    float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy);
    return flow_mix(flow_mix(flow_hash(ix, iy), flow_hash(ix + 1.0f, iy), ux),
                    flow_mix(flow_hash(ix, iy + 1.0f), flow_hash(ix + 1.0f, iy + 1.0f), ux), uy);
}

void orc_flow_direction(const orc_grid *g0, const float *img, const float *sel, float *codes)
{
    orc_grid gm = *g0;                               /* coordsMod (:253-256): float += double literal */
    gm.coords[0] = (float)((double)gm.coords[0] + 1e-3);
    gm.coords[1] = (float)((double)gm.coords[1] - 1e-3);
    const orc_grid *g = &gm;
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        float psx, psy;
        orc_pixelsize(g, sy, &psx, &psy);
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_pixelsize(g, sy, &psx, &psy); }
            float *fc = codes + (size_t)y * W + x;
            float a = fetch_nearest(img, W, H, 1, 0, sx, sy);
            *fc = 0.0f;
            if (a <= 0.0f) continue;
            if (fetch_nearest(sel, W, H, 1, 0, sx, sy) == 0) continue;
            *fc = 5.0f;
            /* get_aspect -> get_texture_gradient (src/Shader.h:281-297, :327-330) */
            float t[9];
            const float ox[9] = { 1, 0, -1, 1, 0, -1, 1, 0, -1 }, oy[9] = { 1, 1, 1, 0, 0, 0, -1, -1, -1 };  /* a..i = -vec2(ox,oy) */
            for (int k = 0; k < 9; k++) {
                float px, py;
                orc_offset(g, sx, sy, -ox[k], -oy[k], rx, ry, &px, &py);
                t[k] = fetch_nearest(img, W, H, 1, 0, px, py);
            }
            float A = t[0], B = t[1], C = t[2], D = t[3], F = t[5], G = t[6], Hh = t[7], I = t[8];
            float gx = (-(C + 2 * F + I) + (A + 2 * D + G)) / (8 * psx);
            float gy = ((G + 2 * Hh + I) - (A + 2 * B + C)) / (8 * psy);
            float aspect = MPI - gatan2(gy, -gx);
            float lower_aspect = floorf(aspect / (2 * MPI) * 8) / 8 * 2 * MPI;
            float upper_aspect = ceilf(aspect / (2 * MPI) * 8) / 8 * 2 * MPI;
            float prob = fabsf(aspect - lower_aspect) / MPI * 4;
            float q = flow_noise(sx * rx * 2, sy * ry * 2) * 0.5f + 0.5f;
            aspect = (q < prob) ? upper_aspect : lower_aspect;
            float dx = roundf(gcos(aspect)), dy = -roundf(gsin(aspect));
            if (dx == 1 && dy == 1) *fc = 9;
            if (dx == 0 && dy == 1) *fc = 8;
            if (dx == -1 && dy == 1) *fc = 7;
            if (dx == 1 && dy == 0) *fc = 6;
            if (dx == -1 && dy == 0) *fc = 4;
            if (dx == 1 && dy == -1) *fc = 3;
            if (dx == 0 && dy == -1) *fc = 2;
            if (dx == -1 && dy == -1) *fc = 1;
            float px, py;
            orc_offset(g, sx, sy, dx, dy, rx, ry, &px, &py);
            float a2 = fetch_nearest(img, W, H, 1, 0, px, py);
            float s2 = fetch_nearest(sel, W, H, 1, 0, px, py);
            if (a2 <= 0.0f) *fc = 5;
            if (s2 == 0) *fc = 5;
            if (a2 < a) continue;
            *fc = 5;
            float s = 0.0f;                               /* GLSL leaves `s` undefined when no neighbour is lower; see note */
            int any = 0;
            const float nx[8] = { 1, 0, -1, 1, -1, 1, 0, -1 }, ny[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };
            const float nc[8] = { 9, 8, 7, 6, 4, 3, 2, 1 };
            for (int k = 0; k < 8; k++) {
                orc_offset(g, sx, sy, nx[k], ny[k], rx, ry, &px, &py);
                a2 = fetch_nearest(img, W, H, 1, 0, px, py);
                s2 = fetch_nearest(sel, W, H, 1, 0, px, py);
                if (a2 < a) { *fc = nc[k]; a = a2; s = s2; any = 1; }
            }
            if (a <= 0.0f) *fc = 5;                       /* edge of ocean */
            if (any && s == 0) *fc = 5;                   /* `if(s==0) fc=5` with s set; unset s only when fc is already 5 */
        }
    }
}

void orc_flow_mask(const orc_grid *g, const float *img, const float *sel, const float *codes, float *mask)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    const float nx[8] = { 1, 0, -1, 1, -1, 1, 0, -1 }, ny[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };
    const float want[8] = { 1, 2, 3, 4, 6, 7, 8, 9 }, add[8] = { 256, 128, 64, 32, 8, 4, 2, 1 };
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) st_raster(g, x, y, &sx, &sy);
            float *fc = mask + (size_t)y * W + x;
            *fc = -1.0f;
            if (fetch_nearest(img, W, H, 1, 0, sx, sy) <= 0.0f) continue;
            if (fetch_nearest(sel, W, H, 1, 0, sx, sy) == 0) continue;
            float v = 0.5f;
            int flag = 0;
            for (int k = 0; k < 8; k++) {
                float px, py;
                orc_offset(g, sx, sy, nx[k], ny[k], rx, ry, &px, &py);
                float c = fetch_nearest(codes, W, H, 1, 0, px, py);
                if (c == want[k]) v += add[k];
                if (c == 0) flag = 1;
            }
            if (fetch_nearest(codes, W, H, 1, 0, sx, sy) == 5) v += 16;
            /* the reference adds in the order 256,128,64,32,16,8,4,2,1: all exact in fp32, order irrelevant */
            if (flag) v += 512;
            *fc = v;
        }
    }
}

void orc_lake_adjacency(const orc_grid *g, const float *lakeid, float *out)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    const float nx[8] = { 1, 0, -1, 1, -1, 1, 0, -1 }, ny[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };
    const float add[8] = { 256, 128, 64, 32, 8, 4, 2, 1 };
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y);
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            float a = fetch_nearest(lakeid, W, H, 1, 0, sx, sy);
            float v = 0.0f;
            if (!(a < 0.0f)) {
                for (int k = 0; k < 8; k++) {
                    float px, py;
                    orc_offset(g, sx, sy, nx[k], ny[k], rx, ry, &px, &py);
                    float a2 = fetch_nearest(lakeid, W, H, 1, 0, px, py);
                    if (a2 != a && a2 > 0) v += add[k];
                }
            }
            out[(size_t)y * W + x] = v;
        }
    }
}

/* ====================================================================== */
/* ThermalErosion and Morphological                        */
/* ====================================================================== */
/* get_slope(1, st) (src/Shader.h:338-341) over get_texture_gradient (:281-297) */
static float orc_slope(const orc_grid *g, const float *img, float sx, float sy, float psx, float psy)
{
    const int W = g->W, H = g->H;
    float t[9];
    const float ox[9] = { 1, 0, -1, 1, 0, -1, 1, 0, -1 }, oy[9] = { 1, 1, 1, 0, 0, 0, -1, -1, -1 };   /* a..i = -vec2(ox,oy) */
    for (int k = 0; k < 9; k++) {
        float px, py;
        orc_offset(g, sx, sy, -ox[k], -oy[k], (float)W, (float)H, &px, &py);
        t[k] = fetch_nearest(img, W, H, 1, 0, px, py);
    }
    float A = t[0], B = t[1], C = t[2], D = t[3], F = t[5], G = t[6], Hh = t[7], I = t[8];
    float gx = (-(C + 2 * F + I) + (A + 2 * D + G)) / (8 * psx);
    float gy = ((G + 2 * Hh + I) - (A + 2 * B + C)) / (8 * psy);
    return gatan(1.0f * sqrtf(gx * gx + gy * gy));
}

/* ThermalErosion::step (src/filter/ThermalErosion.cpp:25-142): 10 x { loss pass :30-72, gain pass :84-128 } in place */
void orc_thermal_erosion(const orc_grid *g, float *h, float *scratch, float *scratch2)
{
    const int W = g->W, H = g->H;
    const float rx = (float)W, ry = (float)H;
    float *img = h, *loss = scratch, *out = scratch2;
    static const float l1x[8] = { -1, 0, 1, 1, -1, 1, 0, -1 }, l1y[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };     /* :41-63 */
    static const float l2x[8] = { -1, 0, 1, -1, 1, -1, 0, 1 }, l2y[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };     /* :92-113 */
    for (int it = 0; it < 10; it++) {
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
        for (int y = 0; y < H; y++) {
            float sy = sty(g, y), psx, psy;
            orc_pixelsize(g, sy, &psx, &psy);
            for (int x = 0; x < W; x++) {
                float sx = stx(g, x);
                if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_pixelsize(g, sy, &psx, &psy); }
                float slope = orc_slope(g, img, sx, sy, psx, psy);
                float hh = fetch_nearest(img, W, H, 1, 0, sx, sy);
                float fc = 0;
                if (slope > MPI / 6 && hh > 0) {
                    float count = 1, minh = hh;
                    for (int k = 0; k < 8; k++) {
                        float px, py;
                        orc_offset(g, sx, sy, l1x[k], l1y[k], rx, ry, &px, &py);
                        float h2 = fetch_nearest(img, W, H, 1, 0, px, py);
                        if (h2 < minh) minh = h2;
                        if (h2 < hh) count += 1;
                    }
                    fc = (hh - minh) / count * 0.3f;
                }
                loss[(size_t)y * W + x] = fc;
            }
        }
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
        for (int y = 0; y < H; y++) {
            float sy = sty(g, y), psx, psy;
            orc_pixelsize(g, sy, &psx, &psy);
            for (int x = 0; x < W; x++) {
                float sx = stx(g, x);
                if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_pixelsize(g, sy, &psx, &psy); }
                float hh = fetch_nearest(img, W, H, 1, 0, sx, sy);
                float gain = 0;
                for (int k = 0; k < 8; k++) {
                    float px, py;
                    orc_offset(g, sx, sy, l2x[k], l2y[k], rx, ry, &px, &py);
                    float h2 = fetch_nearest(img, W, H, 1, 0, px, py);
                    if (h2 > hh) gain += fetch_nearest(loss, W, H, 1, 0, px, py);
                }
                float slope = orc_slope(g, img, sx, sy, psx, psy);
                out[(size_t)y * W + x] = (slope > MPI / 6 / 10 && hh > 0) ? hh : hh + gain;      /* :121-125 */
            }
        }
        float *t = img; img = out; out = t;          /* scratch2.swap(terrain) (:134) */
    }
    if (img != h) memcpy(h, img, sizeof(float) * (size_t)W * H);
}

/* Morphological radius schedule (src/filter/Morphological.cpp:31-43) */
int orc_morph_rlist(float radius, float *r, int cap)
{
    int n = 0, x = 1;
    while (radius >= 0) {
        if (x < radius) { radius -= x; if (n < cap) r[n] = (float)x; n++; x *= 2; }
        else { if (n < cap) r[n] = radius; n++; break; }
    }
    int m = n < cap ? n : cap;
    for (int a = 1; a < m; a++) { float k = r[a]; int b = a - 1; while (b >= 0 && r[b] > k) { r[b + 1] = r[b]; b--; } r[b + 1] = k; }
    return n;
}

/* one Morphological::step (:75-85): 64-tap ring min (use_max = 0) or max (:49-60), NEAREST target */
void orc_morph_pass(const orc_grid *g, const float *src, float radius, int use_max, float *dst)
{
    const int W = g->W, H = g->H;
    float cs[64], sn[64];
    for (int i = 0; i < 64; i++) { float a = 2 * 3.14159f * (float)i / 64.0f; cs[i] = gcos(a); sn[i] = gsin(a); }
#pragma omp parallel for schedule(static) num_threads(nthreads_rows(H, 0))
    for (int y = 0; y < H; y++) {
        float sy = sty(g, y), lon, phi;
        orc_tex_to_spheric(g, 0.0f, sy, &lon, &phi);
        float factor = 1 / gcos(fabsf(phi));
        for (int x = 0; x < W; x++) {
            float sx = stx(g, x);
            if (g_raster_st) { st_raster(g, x, y, &sx, &sy); orc_tex_to_spheric(g, 0.0f, sy, &lon, &phi); factor = 1 / gcos(fabsf(phi)); }
            float a = fetch_nearest(src, W, H, 1, 0, sx, sy);
            for (int i = 0; i < 64; i++) {
                float px, py;
                orc_offset(g, sx, sy, cs[i] * radius * factor, sn[i] * radius, (float)W, (float)H, &px, &py);
                float v = fetch_nearest(src, W, H, 1, 0, px, py);
                a = use_max ? fmaxf(a, v) : fminf(a, v);
            }
            dst[(size_t)y * W + x] = a;
        }
    }
}

void orc_morphological(const orc_grid *g, float *field, float *scratch, float radius, int use_max)
{
    float r[64];
    int n = orc_morph_rlist(radius, r, 64);
    float *a = field, *b = scratch;
    for (int i = 0; i < n; i++) { orc_morph_pass(g, a, r[i], use_max, b); float *t = a; a = b; b = t; }
    if (a != field) memcpy(field, a, sizeof(float) * (size_t)g->W * g->H);
}
