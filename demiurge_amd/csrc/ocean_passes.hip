// The non-Jacobi passes of OceanCurrents::run (reference src/filter/OceanCurrents.cpp:155-209):
// advect (:235-360), diffusion (:625-708), divergence (:363-435), subDiv (:511-622) and the outer step.
// One thread per cell, coalesced row-major access; velocity is interleaved (vx,vy) like the reference's RG32F.
// Arithmetic follows the GLSL statement order with IEEE ops and host-computed per-row tables; GLSL transcendentals
// (advect, subDiv) go through glsl_math.h's restatement of Mesa/gallivm's forms, never through device libm, so every
// pass matches the oracle -- and the reference's llvmpipe output on power-of-two grids -- bit for bit.
#include "ocean_internal.h"

struct pass_args {
    const float2 *vin;
    float2 *vout;
    const float *p;
    float *out;                      // scalar output (divergence)
    const float *img;
    const unsigned char *code;       // bits0-3 N,S,E,W "use centre" (solid or clamped onto self), bit4 centre solid
    const unsigned char *code8;      // bit i: neighbour offsets[i] is solid, offsets = E,NE,N,NW,W,SW,S,SE (:560)
    const float *rt;                 // row tables, ROWTAB_FIELDS x L
    int W, L;
    int l_south_pole, l_north_pole;  // local row of grid row 0 / H-1 when that edge is a pole, else -1
    int lonwrap;
    float pressurefactor, scale, timestep, wind, circumference;
    float c0, c1, c2, c3;            // cornerCoords
    int scalar_rhs;
    int H;                           // full grid height (advect's texture space)
    int y_first;                     // grid row of local row 0
    int row_lo;                      // first local row computed by the launch (blockIdx.y = 0)
    int *oob_flag;                   // advect: set to 1 when a departure point needs rows outside the band's storage
    const float2 *vg;                // advect: the rows the back-trace gathers from (v itself, or the deeper copy of
    int g_y_first, g_L;              //         dmg_ocean_advect_prepare): g_L rows from grid row g_y_first on
    // pixelsize(st_o).x*.y of the virtual rows y = -1 / y = H: on a non-pole edge offset() leaves st_o.y outside
    // [0,1] (only the texture fetch clamps), so get_velocity()'s d_o is evaluated beyond the grid (:380-381,:392)
    float area_below, area_above;
    int l_south_clamp, l_north_clamp;   // local row of grid row 0 / H-1 when that edge is NOT a pole, else -1
};

// resolves the 4-neighbourhood of (x,l): neighbour column/row and whether the tap crossed a pole
struct nb4 {
    int xn, ln, xs, ls, xe, xw;
    bool nflip, sflip;
};
__device__ __forceinline__ nb4 neighbours(const pass_args &a, int x, int l)
{
    nb4 n;
    const int W = a.W;
    n.xe = x + 1; n.xw = x - 1;
    if (n.xe >= W) n.xe = a.lonwrap ? 0 : W - 1;
    if (n.xw < 0) n.xw = a.lonwrap ? W - 1 : 0;
    n.nflip = (l == a.l_north_pole); n.sflip = (l == a.l_south_pole);
    if (n.nflip) { n.ln = l; n.xn = x + W / 2; if (n.xn >= W) n.xn -= W; }
    else { n.ln = (l + 1 < a.L) ? l + 1 : l; n.xn = x; }
    if (n.sflip) { n.ls = l; n.xs = x + W / 2; if (n.xs >= W) n.xs -= W; }
    else { n.ls = (l >= 1) ? l - 1 : l; n.xs = x; }
    return n;
}

__device__ __forceinline__ float2 neg2(float2 v) { return make_float2(-v.x, -v.y); }

// ---------------------------------------------------------------------------------------------
// divergence (:376-420)
// ---------------------------------------------------------------------------------------------
template <bool X>
__global__ __launch_bounds__(256) void k_divergence(pass_args a)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, l = a.row_lo + blockIdx.y;
    if (x >= a.W) return;
    const int W = a.W, L = a.L;
    nb4 n = neighbours(a, x, l);
    const float *dxr = a.rt + (size_t)RT_DX * L, *dyr = a.rt + (size_t)RT_DY * L;
    unsigned c8 = a.code8[(size_t)l * W + x];
    // get_velocity(): v(st_o) [sign-flipped across a pole, :389] * (d_o.x*d_o.y) * pressurefactor (:392)
    float2 vN = a.vin[(size_t)n.ln * W + n.xn], vS = a.vin[(size_t)n.ls * W + n.xs];
    float2 vE = a.vin[(size_t)l * W + n.xe], vW = a.vin[(size_t)l * W + n.xw];
    if (n.nflip) vN = neg2(vN);
    if (n.sflip) vS = neg2(vS);
    float aN = __fmul_rn(dxr[n.ln], dyr[n.ln]), aS = __fmul_rn(dxr[n.ls], dyr[n.ls]), aC = __fmul_rn(dxr[l], dyr[l]);
    if (l == a.l_north_clamp) aN = a.area_above;
    if (l == a.l_south_clamp) aS = a.area_below;
    float vNy = __fmul_rn(__fmul_rn(vN.y, aN), a.pressurefactor);
    float vSy = __fmul_rn(__fmul_rn(vS.y, aS), a.pressurefactor);
    float vEx = __fmul_rn(__fmul_rn(vE.x, aC), a.pressurefactor);
    float vWx = __fmul_rn(__fmul_rn(vW.x, aC), a.pressurefactor);
    if (c8 & 4u) vNy = 0.0f;       // oN > 0 (:413)
    if (c8 & 64u) vSy = 0.0f;
    if (c8 & 1u) vEx = 0.0f;
    if (c8 & 16u) vWx = 0.0f;
    if (X) {
        float pwx = __fdiv_rn(dxr[l], a.scale), pwy = __fdiv_rn(dyr[l], a.scale);    // pixelsize(st)/420 (:419)
        a.out[(size_t)l * W + x] =
            __fmul_rn(0.5f, __fadd_rn(__fdiv_rn(__fsub_rn(vEx, vWx), pwx), __fdiv_rn(__fsub_rn(vNy, vSy), pwy)));   // :420
    } else {
        // the row's two quotients as reciprocals (wave-uniform: one v_rcp each), the per-cell divisions as multiplications
        const float rpwx = a.scale * __frcp_rn(dxr[l]), rpwy = a.scale * __frcp_rn(dyr[l]);
        a.out[(size_t)l * W + x] = 0.5f * ((vEx - vWx) * rpwx + (vNy - vSy) * rpwy);
    }
}

// ---------------------------------------------------------------------------------------------
// velocity diffusion sweep (:664-693)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_diffusion(pass_args a)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, l = a.row_lo + blockIdx.y;
    if (x >= a.W) return;
    const int W = a.W, L = a.L;
    nb4 n = neighbours(a, x, l);
    unsigned c = a.code[(size_t)l * W + x];
    float2 pC = a.vin[(size_t)l * W + x];
    float2 pN = a.vin[(size_t)n.ln * W + n.xn], pS = a.vin[(size_t)n.ls * W + n.xs];
    float2 pE = a.vin[(size_t)l * W + n.xe], pW = a.vin[(size_t)l * W + n.xw];
    if (n.nflip) pN = neg2(pN);      // :657
    if (n.sflip) pS = neg2(pS);
    if (c & 1u) pN = pC;             // :682-685
    if (c & 2u) pS = pC;
    if (c & 4u) pE = pC;
    if (c & 8u) pW = pC;
    float px = a.rt[(size_t)RT_DIFPX * L + l], py = a.rt[(size_t)RT_DIFPY * L + l], beta = a.rt[(size_t)RT_DIFBETA * L + l];
    float bx = pC.x, by = a.scalar_rhs ? pC.x : pC.y;      // `float bC = texture(v,st).r` (:687)
    float2 o;
    // ((pW + pE)*pw2.x + (pS + pN)*pw2.y - (-bC)) / Beta   (:693); Mesa associates the sum as (bC + a) + b
    o.x = __fdiv_rn(__fadd_rn(__fadd_rn(bx, __fmul_rn(__fadd_rn(pW.x, pE.x), px)), __fmul_rn(__fadd_rn(pS.x, pN.x), py)), beta);
    o.y = __fdiv_rn(__fadd_rn(__fadd_rn(by, __fmul_rn(__fadd_rn(pW.y, pE.y), px)), __fmul_rn(__fadd_rn(pS.y, pN.y), py)), beta);
    a.vout[(size_t)l * W + x] = o;
}

// ---------------------------------------------------------------------------------------------
// subDiv: pressure-gradient subtraction + coastline direction snapping (:532-606)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float glsl_mod(float x, float y) { return __fsub_rn(x, __fmul_rn(y, floorf(__fdiv_rn(x, y)))); }

__global__ __launch_bounds__(256) void k_subdiv(pass_args a)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, l = a.row_lo + blockIdx.y;
    if (x >= a.W) return;
    const int W = a.W, L = a.L;
    size_t i = (size_t)l * W + x;
    unsigned c = a.code[i];
    if (c & 16u) { a.vout[i] = make_float2(0.0f, 0.0f); return; }     // :536-540
    nb4 n = neighbours(a, x, l);
    float pC = a.p[i];
    float pN = a.p[(size_t)n.ln * W + n.xn], pS = a.p[(size_t)n.ls * W + n.xs];
    float pE = a.p[(size_t)l * W + n.xe], pW = a.p[(size_t)l * W + n.xw];
    if (c & 1u) pN = pC;
    if (c & 2u) pS = pC;
    if (c & 4u) pE = pC;
    if (c & 8u) pW = pC;
    float2 oldV = a.vin[i];
    float dx = a.rt[(size_t)RT_DX * L + l], dy = a.rt[(size_t)RT_DY * L + l];
    float pwx = __fdiv_rn(dx, a.scale), pwy = __fdiv_rn(dy, a.scale);
    // :575  oldV - 0.5*vec2(pE-pW,pN-pS)/pixelwidth / pixelsize.x / pixelsize.y / pressurefactor
    float fx = __fsub_rn(oldV.x, __fdiv_rn(__fdiv_rn(__fdiv_rn(__fdiv_rn(__fmul_rn(0.5f, __fsub_rn(pE, pW)), pwx), dx), dy), a.pressurefactor));
    float fy = __fsub_rn(oldV.y, __fdiv_rn(__fdiv_rn(__fdiv_rn(__fdiv_rn(__fmul_rn(0.5f, __fsub_rn(pN, pS)), pwy), dx), dy), a.pressurefactor));
    unsigned o8 = a.code8[i];                                             // o_array[k] > 0  <=>  bit k
    float ang = __fadd_rn(__fmul_rn(__fdiv_rn(__fadd_rn(__fdiv_rn(glsl::atan2(fy, fx), DMG_PI), 1.0f), 2.0f), 8.0f), 4.0f);
    float lower = glsl_mod(floorf(ang), 8.0f), upper = glsl_mod(ceilf(ang), 8.0f), theta = glsl_mod(ang, 8.0f);
    int il = min(max((int)lower, 0), 7), iu = min(max((int)upper, 0), 7);
    bool isBorder = ((o8 >> il) & 1u) || ((o8 >> iu) & 1u);
    float nx = fx, ny = fy;
    float difference = 2.0f * DMG_PI;
    const float offx[8] = { 1, 1, 0, -1, -1, -1, 0, 1 }, offy[8] = { 0, 1, 1, 1, 0, -1, -1, -1 };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // thetai = mod((atan(o.y,o.x)/M_PI+1)/2*8+4, 8) is exactly k for the 8 compass offsets (:597)
        float thetai = (float)k;
        float d = fabsf(__fsub_rn(thetai, theta));
        float angleToVel = fminf(__fsub_rn(2.0f * DMG_PI, d), d);          // sic: 2*pi, not 8 (:598)
        if (angleToVel < difference && !((o8 >> k) & 1u)) {
            difference = angleToVel;
            float lo = sqrtf(__fadd_rn(__fmul_rn(offx[k], offx[k]), __fmul_rn(offy[k], offy[k])));
            float ln = sqrtf(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)));
            nx = __fmul_rn(__fdiv_rn(offx[k], lo), ln);
            ny = __fmul_rn(__fdiv_rn(offy[k], lo), ln);
        }
    }
    if (isBorder) { fx = nx; fy = ny; }
    a.vout[i] = make_float2(fx, fy);
}

// ---------------------------------------------------------------------------------------------
// advect: semi-Lagrangian back-trace along a great circle + parallel transport + wind forcing (:282-348)
// ---------------------------------------------------------------------------------------------
struct vec3 { float x, y, z; };
__device__ __forceinline__ vec3 v3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ vec3 cross3(vec3 a, vec3 b)
{ return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
__device__ __forceinline__ float dot3(vec3 a, vec3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }   // Mesa's dot(): from the last component down

// The transcendental and division steps of the trig-heavy passes, in two renderings (dmg_ocean_params.quirks,
// DMG_QUIRK_FAST_TRANSCENDENTALS set: the second).  EXACT: glsl_math.h's restatement of Mesa / gallivm, IEEE divisions and square roots -- the
// reference's pixels bit for bit, at ~530 vector instructions per advected cell.  Otherwise the SAME expressions (Mesa's asin
// and atan polynomials are approximations with 1e-4 rad of error of their own: a "better" arcsine would be further from the
// reference than this) on the hardware's sine, cosine, log2, exp2, reciprocal and reciprocal square root (1 ulp each):
// within north_star's 1e-5 of the exact rendering (tests/test_gpu_ocean_passes.py), a third of the instructions.
template <bool EXACT> struct gmath {
    // (sine and cosine stay gallivm's octant scheme in both renderings: the hardware's v_sin / v_cos are good to ~1e-6 ABSOLUTE, and
    // a longitude that is 1e-6 rad off is 2.6e-3 cells off on a 16384-wide grid -- measured on the 128x64 golden: 3.0e-5 of the
    // field's maximum with them, inside 1e-5 without.  What the tolerance rendering drops is the IEEE division and square-root
    // sequences and Mesa's log2 / exp2 polynomials.)
    static __device__ __forceinline__ float sin(float x) { return glsl::sin(x); }
    static __device__ __forceinline__ float cos(float x) { return glsl::cos(x); }
    static __device__ __forceinline__ float rcp(float x) { return EXACT ? 1.0f / x : __frcp_rn(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return EXACT ? 1.0f / sqrtf(x) : __frsqrt_rn(x); }
    static __device__ __forceinline__ float sqrt(float x) { return EXACT ? sqrtf(x) : __fsqrt_rn(x); }
    static __device__ __forceinline__ float div(float a, float b) { return EXACT ? a / b : __fdividef(a, b); }
    static __device__ __forceinline__ float pow(float x, float y) { return EXACT ? glsl::pow(x, y) : __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * y); }
    static __device__ __forceinline__ float asin(float x)
    {
        if (EXACT) return glsl::asin(x);
        const float ax = fabsf(x);
        float t = 0.086566724f + ax * -0.03102955f;
        t = (glsl::PI_4 - 1.0f) + ax * t;
        t = glsl::PI_2 + ax * t;
        return glsl::sign(x) * (glsl::PI_2 - __fsqrt_rn(1.0f - ax) * t);
    }
    static __device__ __forceinline__ float atan_pos(float a)
    {
        const float x = fminf(a, 1.0f) * __frcp_rn(fmaxf(a, 1.0f));
        const float x2 = x * x, x3 = x2 * x, x5 = x3 * x2, x7 = x5 * x2, x9 = x7 * x2;
        float p = x * 0.9999793128310355f;
        p = p + x3 * -0.3326756418091246f;
        p = p + x5 * 0.1938924977115610f;
        p = p + x7 * -0.1173503194786851f;
        p = p + x9 * 0.0536813784310406f;
        p = p + (x9 * -0.0121323213173444f) * x2;
        return p + (a > 1.0f ? 1.0f : 0.0f) * (p * -2.0f + glsl::PI_2);
    }
    static __device__ __forceinline__ float atan2(float y, float x)
    {
        if (EXACT) return glsl::atan2(y, x);
        const bool flip = 0.0f >= x;
        const float s = flip ? fabsf(x) : y, t = flip ? y : fabsf(x);
        const float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
        const float rcp_scaled_t = __frcp_rn(t * scale);
        const float s_over_t = (s * scale) * rcp_scaled_t;
        const float tn = fabsf(x) == fabsf(y) ? 1.0f : fabsf(s_over_t);
        const float arc = (flip ? 1.0f : 0.0f) * glsl::PI_2 + atan_pos(tn) * glsl::sign(tn);
        return fminf(y, rcp_scaled_t) < 0.0f ? -arc : arc;
    }
};
template <bool X> __device__ __forceinline__ vec3 norm3(vec3 a) { float r = gmath<X>::rsqrt(dot3(a, a)); return v3(a.x * r, a.y * r, a.z * r); }
template <bool X> __device__ __forceinline__ vec3 sph2cart(float lon, float lat)
{
    const float cl = gmath<X>::cos(lat);
    return v3(cl * gmath<X>::cos(lon), cl * gmath<X>::sin(lon), gmath<X>::sin(lat));
}
// v_to_cartesian / cartesian_to_v (:251-268)
template <bool X> __device__ __forceinline__ void tangent_basis(float lon, float lat, vec3 *par, vec3 *ycomp)
{
    vec3 c = sph2cart<X>(lon, lat);
    float il = gmath<X>::rsqrt(c.y * c.y + c.x * c.x);      // normalize(c.xy): dot() from the last component down
    float inx = c.x * il, iny = c.y * il;
    *ycomp = v3(gmath<X>::sin(lat) * (-inx), gmath<X>::sin(lat) * (-iny), gmath<X>::cos(lat));
    *par = norm3<X>(cross3(v3(0, 0, 1), c));
}
// rotation_matrix(theta,u)*v (reference src/Shader.h:35-40; mat3 columns)
template <bool X> __device__ __forceinline__ vec3 rotate3(float theta, vec3 u, vec3 v)
{
    float c = gmath<X>::cos(theta), s = gmath<X>::sin(theta), ic = 1.0f - c;
    vec3 c0 = v3(c + u.x * u.x * ic, u.y * u.x * ic + u.z * s, u.z * u.x * ic - u.y * s);
    vec3 c1 = v3(u.x * u.y * ic - u.z * s, c + u.y * u.y * ic, u.z * u.y * ic + u.x * s);
    vec3 c2 = v3(u.x * u.z * ic + u.y * s, u.y * u.z * ic - u.x * s, c + u.z * u.z * ic);
    return v3(c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z, c0.z * v.x + c1.z * v.y + c2.z * v.z);
}
// GL_LINEAR + GL_CLAMP_TO_EDGE fetch of the RG32F velocity texture at (u,v) (reference src/Texture.cpp:29-33).
// (u,v) address the FULL grid; `t` holds the band's L rows starting at grid row y_first.  A tap outside the stored
// rows raises *oob (the caller needs more ghost rows) and is clamped so that nothing is read out of bounds.
__device__ __forceinline__ float2 fetch_linear2(const float2 *t, int W, int H, float u, float v, int y_first, int L, int *oob)
{
    float fu = u * (float)W - 0.5f, fv = v * (float)H - 0.5f;
    float flu = floorf(fu), flv = floorf(fv);
    float wx = fu - flu, wy = fv - flv;
    int i0 = min(max((int)flu, 0), W - 1), i1 = min(max((int)flu + 1, 0), W - 1);
    int j0 = min(max((int)flv, 0), H - 1) - y_first, j1 = min(max((int)flv + 1, 0), H - 1) - y_first;
    if (j0 < 0 || j1 >= L) {
        // a cell at rest has no back-trace direction (normalize(0), :306): its coordinates are NaN and whatever is fetched
        // is discarded (:321) -- that is not a reach beyond the band
        if (fv == fv && fu == fu) *oob = 1;
        j0 = min(max(j0, 0), L - 1); j1 = min(max(j1, 0), L - 1);
    }
    float2 t00 = t[(size_t)j0 * W + i0], t10 = t[(size_t)j0 * W + i1], t01 = t[(size_t)j1 * W + i0], t11 = t[(size_t)j1 * W + i1];
    // the reference's pixels come from a filter whose lerp is one fused multiply-add, a + w*(b-a) (llvmpipe lp_build_lerp)
    float ax = fmaf(wx, t10.x - t00.x, t00.x), bx = fmaf(wx, t11.x - t01.x, t01.x);
    float ay = fmaf(wx, t10.y - t00.y, t00.y), by = fmaf(wx, t11.y - t01.y, t01.y);
    return make_float2(fmaf(wy, bx - ax, ax), fmaf(wy, by - ay, ay));
}

template <bool X>
__global__ __launch_bounds__(256) void k_advect(pass_args a, float k04)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, l = a.row_lo + blockIdx.y;
    if (x >= a.W) return;
    const int W = a.W;
    size_t i = (size_t)l * W + x;
    if (a.code[i] & 16u) { a.vout[i] = make_float2(0.0f, 0.0f); return; }        // :284-288
    float2 vel = a.vin[i];                                                         // :290
    float sx = gmath<X>::div((float)x + 0.5f, (float)W), sy = gmath<X>::div((float)(a.y_first + l) + 0.5f, (float)a.H);
    float distance = gmath<X>::sqrt(vel.x * vel.x + vel.y * vel.y) * a.timestep;           // :298
    float arclength = gmath<X>::div(2.0f * 3.14159f, a.circumference) * distance;                // :299
    float lon = sx * (a.c3 - a.c2) + a.c2, lat = sy * (a.c1 - a.c0) + a.c0;        // tex_to_spheric
    vec3 cart = sph2cart<X>(lon, lat);
    vec3 par, yc;
    tangent_basis<X>(lon, lat, &par, &yc);
    vec3 v_cart = v3(par.x * vel.x + yc.x * vel.y, par.y * vel.x + yc.y * vel.y, par.z * vel.x + yc.z * vel.y);
    vec3 rotdir = norm3<X>(cross3(cart, v_cart));                                     // :306
    cart = rotate3<X>(-arclength, rotdir, cart);                                      // :308
    float lon_n = gmath<X>::atan2(cart.y, cart.x), lat_n = gmath<X>::asin(cart.z);                   // cartesian_to_spheric
    float cu = gmath<X>::div(lon_n - a.c2, a.c3 - a.c2), cv = gmath<X>::div(lat_n - a.c0, a.c1 - a.c0);   // spheric_to_tex
    float2 nv = fetch_linear2(a.vg, W, a.H, cu, cv, a.g_y_first, a.g_L, a.oob_flag);  // :315
    vec3 par_n, yc_n;
    tangent_basis<X>(lon_n, lat_n, &par_n, &yc_n);
    vec3 testv = v3(par_n.x * nv.x + yc_n.x * nv.y, par_n.y * nv.x + yc_n.y * nv.y, par_n.z * nv.x + yc_n.z * nv.y);
    testv = rotate3<X>(arclength, rotdir, testv);
    float nvx = dot3(testv, par), nvy = dot3(testv, yc);                           // cartesian_to_v(testv, tex_to_spheric(st))
    if (isnan(nvx) || isnan(nvy)) { nvx = 0.0f; nvy = 0.0f; }                      // :321 (zero velocity: normalize(0))
    float fcx = 1.0f * nvx, fcy = 1.0f * nvy;                                      // Dissipation :332; Coriolis x0.0 (:327)
    float phi = 2.0f * (sy - 0.5f) * 3.14159f;                                     // :341
    // phi*3/2: Mesa merges the literal factors of the product chain into one constant (pinned with a probe shader)
    const float phi32 = (sy - 0.5f) * ((2.0f * 3.14159f) * (3.0f * 0.5f));
    float windx = a.wind * -gmath<X>::cos(phi32), windy = a.wind * gmath<X>::sin(phi32);
    if (fabsf(phi32) > 3.14159f) windx = -windx;
    if (fabsf(phi) > 3.14159f * 1.0f / 3.0f && fabsf(phi) < 3.1459f * 2.0f / 3.0f) windy = -windy;   // sic 3.1459 (:344)
    if (phi < 0.0f) windy = -windy;
    float ax = fabsf(windx - fcx), ay = fabsf(windy - fcy);
    float stressx = 1.0f + (0.0001f * ax) * ax, stressy = 1.0f + (0.0001f * ay) * ay;   // :347, pow(x,2) -> x*x, associated as Mesa does
    float e = -1.0f / 24.0f * 2.0f;
    float2 o;
    o.x = fcx + windx * (1.0f - gmath<X>::pow(stressx, e)) - fcx * (1.0f - k04);            // :348
    o.y = fcy + windy * (1.0f - gmath<X>::pow(stressy, e)) - fcy * (1.0f - k04);
    a.vout[i] = o;
}

// neighbour-solid bits for the 8 compass offsets of subDiv (:560), img fetched through offset() + CLAMP_TO_EDGE
__global__ void k_build_codes8(const float *__restrict__ img, unsigned char *__restrict__ code8, int W, int L,
                               int y_first, int H, int lonwrap, int spole, int npole)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (x >= W) return;
    const int offx[8] = { 1, 1, 0, -1, -1, -1, 0, 1 }, offy[8] = { 0, 1, 1, 1, 0, -1, -1, -1 };
    int y = y_first + l;
    unsigned c = 0;
    for (int k = 0; k < 8; k++) {
        int xx = x + offx[k], yy = y + offy[k];
        // offset(): longitude wrap first (Shader.h:85), then pole reflection with lon += pi (:86-95), else clamp
        if (lonwrap) { if (xx >= W) xx -= W; if (xx < 0) xx += W; }
        else xx = min(max(xx, 0), W - 1);
        if (yy < 0) { if (spole) { yy = 0; xx = (xx + W / 2) % W; } else yy = 0; }
        if (yy >= H) { if (npole) { yy = H - 1; xx = (xx + W / 2) % W; } else yy = H - 1; }
        int ll = yy - y_first;
        bool solid = (ll >= 0 && ll < L) ? (img[(size_t)ll * W + xx] > 0.0f) : true;   // beyond the ghost rows: never used
        if (solid) c |= (1u << k);
    }
    code8[(size_t)l * W + x] = (unsigned char)c;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static pass_args make_args(dmg_ocean *o)
{
    pass_args a;
    memset(&a, 0, sizeof a);
    a.vin = (const float2 *)o->v[o->vcur];
    a.vout = (float2 *)o->v[o->vcur ^ 1];
    a.p = o->p[o->cur];
    a.out = o->divw;
    a.img = o->img;
    a.code = o->code;
    a.code8 = o->code8;
    a.rt = o->rowtab;
    a.W = o->grid.W; a.L = o->L; a.H = o->grid.H; a.y_first = o->y0 - o->gs;
    a.l_south_pole = (o->topo.spole && o->y0 == 0) ? o->gs : -1;
    a.l_north_pole = (o->topo.npole && o->y1 == o->grid.H) ? o->gs + (o->y1 - o->y0) - 1 : -1;
    a.lonwrap = o->topo.lonwrap ? 1 : 0;
    a.pressurefactor = o->prm.pressurefactor; a.scale = o->prm.scale; a.timestep = o->prm.timestep_h;
    a.wind = o->prm.wind; a.circumference = o->grid.circumference;
    a.c0 = o->grid.coords[0]; a.c1 = o->grid.coords[1]; a.c2 = o->grid.coords[2]; a.c3 = o->grid.coords[3];
    a.scalar_rhs = (o->prm.quirks & DMG_QUIRK_DIFFUSION_SCALAR_RHS) ? 1 : 0;
    a.oob_flag = o->oob_flag;
    a.vg = a.vin; a.g_y_first = a.y_first; a.g_L = a.L;
    a.l_south_clamp = (!o->topo.spole && o->y0 == 0) ? o->gs : -1;
    a.l_north_clamp = (!o->topo.npole && o->y1 == o->grid.H) ? o->gs + (o->y1 - o->y0) - 1 : -1;
    float dx, dy;
    dmg_row_pixelsize(o->grid, -1, &dx, &dy);
    a.area_below = dx * dy;
    dmg_row_pixelsize(o->grid, o->grid.H, &dx, &dy);
    a.area_above = dx * dy;
    return a;
}

static int require_terrain(dmg_ocean *o, const char *what)
{
    if (!o) return dmg_fail(DMG_EINVAL, "%s: null handle", what);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "%s: dmg_ocean_set_terrain must come first", what);
    DMG_HIP(hipSetDevice(o->ctx->device));
    return DMG_OK;
}

template <typename K, typename... A>
static int launch_rows(dmg_ocean *o, K kernel, pass_args a, int lo, int hi, A... extra)
{
    if (hi <= lo) return DMG_OK;
    a.row_lo = lo;
    dim3 grid((a.W + 255) / 256, hi - lo);
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, o->ctx->stream, a, extra...);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

int ocean_build_codes8(dmg_ocean *o)
{
    dim3 grid((o->grid.W + 255) / 256, o->L);
    hipLaunchKernelGGL(k_build_codes8, grid, dim3(256), 0, o->ctx->stream, o->img, o->code8, o->grid.W, o->L,
                       o->y0 - o->gs, o->grid.H, (int)o->topo.lonwrap, (int)o->topo.spole, (int)o->topo.npole);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

// max |v|^2 over the owned rows (non-negative floats order like their bit patterns)
__global__ void k_max_speed2(const float2 *__restrict__ v, size_t n, unsigned *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 a = v[i]; m = fmaxf(m, a.x * a.x + a.y * a.y); }
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

extern "C" {

// Rows of velocity halo the next dmg_ocean_advect of THIS band needs: the back-trace travels |v|*timestep km
// (OceanCurrents.cpp:298-299) = that many / (circumference/(2H)) rows (H rows span half the circumference on a
// full globe; other extents scale by their latitude range), plus the bilinear footprint.
int dmg_ocean_advect_reach(dmg_ocean *o, int *rows)
{
    DMG_CHECK_ARG(o != nullptr && rows != nullptr);
    DMG_HIP(hipSetDevice(o->ctx->device));
    DMG_HIP(hipMemsetAsync(o->oob_flag, 0, sizeof(int), o->ctx->stream));
    size_t n = (size_t)(o->y1 - o->y0) * o->grid.W;
    hipLaunchKernelGGL(k_max_speed2, dim3(1024), dim3(256), 0, o->ctx->stream,
                       (const float2 *)o->v[o->vcur] + (size_t)o->gs * o->grid.W, n, (unsigned *)o->oob_flag);
    DMG_HIP(hipGetLastError());
    unsigned bits = 0;
    DMG_HIP(hipMemcpyAsync(&bits, o->oob_flag, sizeof bits, hipMemcpyDeviceToHost, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    float m2;
    memcpy(&m2, &bits, sizeof m2);
    double km = sqrt((double)m2) * o->prm.timestep_h;
    double row_km = (double)(o->grid.coords[1] - o->grid.coords[0]) / (2.0 * M_PI) * o->grid.circumference / o->grid.H;
    *rows = (int)ceil(km / row_km) + 2;
    return DMG_OK;
}

int dmg_ocean_set_velocity(dmg_ocean *o, const float *vx, const float *vy)
{
    DMG_CHECK_ARG(o != nullptr && vx != nullptr && vy != nullptr);
    size_t n = (size_t)o->L * o->grid.W, off = (size_t)(o->y0 - o->gs) * o->grid.W;
    std::vector<float> tmp(n * 2);
    for (size_t i = 0; i < n; i++) { tmp[2 * i] = vx[off + i]; tmp[2 * i + 1] = vy[off + i]; }
    DMG_HIP(hipMemcpyAsync(o->v[o->vcur], tmp.data(), n * 8, hipMemcpyHostToDevice, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    return DMG_OK;
}

// Band semantics of the passes (band_count > 1): a pass computes every local row whose inputs are valid, given that
// the host has refreshed the ghost rows of the input field(s) to full depth (dmg_ocean_halo) beforehand:
//   advect      owned rows                (gathers v anywhere inside the stored rows; fails if it needs more)
//   diffusion   n <= ghost sweeps, region shrinking one row per sweep like dmg_ocean_jacobi_sweeps
//   divergence  local rows [1, L-1) on interior sides, so divw is valid ghost-1 rows deep for the Jacobi halo
//   subDiv      owned rows                (needs 1 row of pressure halo)
int dmg_ocean_divergence(dmg_ocean *o)
{
    int rc = require_terrain(o, "dmg_ocean_divergence");
    if (rc) return rc;
    const bool exact = (o->prm.quirks & DMG_QUIRK_FAST_TRANSCENDENTALS) == 0;
    return exact ? launch_rows(o, k_divergence<true>, make_args(o), o->gs ? 1 : 0, o->L - (o->gn ? 1 : 0))
                 : launch_rows(o, k_divergence<false>, make_args(o), o->gs ? 1 : 0, o->L - (o->gn ? 1 : 0));
}

int dmg_ocean_diffusion(dmg_ocean *o, int iters)
{
    int rc = require_terrain(o, "dmg_ocean_diffusion");
    if (rc) return rc;
    DMG_CHECK_ARG(iters >= 0);
    if (o->count > 1 && iters > o->ghost)
        return dmg_fail(DMG_EINVAL, "%d diffusion sweeps between halo refreshes but only %d ghost rows", iters, o->ghost);
    // after k sweeps since the last halo refresh the rows still computable lie k rows inside the ghost slabs
    static const bool single = dmg_hook("DMG_DIFFUSION_SINGLE") != nullptr;     // experiments: one sweep per launch
    if (!single && o->prm.jacobi_kernel != DMG_JACOBI_SINGLE && ocean_tb_supported(o) && o->rowclass) {
        for (int k = 0; k < iters;) {                     // up to 8 sweeps per launch (ocean_diffusion_tb.hip)
            int m = iters - k < 8 ? iters - k : 8;
            k += m;
            rc = ocean_launch_diffusion_tb(o, o->gs ? k : 0, o->L - (o->gn ? k : 0), m);
            if (rc) return rc;
            o->vcur ^= 1;
        }
        return DMG_OK;
    }
    for (int k = 1; k <= iters; k++) {                    // :701-707
        rc = launch_rows(o, k_diffusion, make_args(o), o->gs ? k : 0, o->L - (o->gn ? k : 0));
        if (rc) return rc;
        o->vcur ^= 1;                                      // v_scratch->swap(v)
    }
    return DMG_OK;
}

int dmg_ocean_subdiv(dmg_ocean *o)
{
    int rc = require_terrain(o, "dmg_ocean_subdiv");
    if (rc) return rc;
    if ((rc = no_pending(o, "dmg_ocean_subdiv"))) return rc;
    rc = launch_rows(o, k_subdiv, make_args(o), o->gs, o->gs + (o->y1 - o->y0));
    if (rc) return rc;
    o->vcur ^= 1;                                          // v->swap(v_scratch) (:621)
    return DMG_OK;
}

// A back-trace longer than the band's `ghost` rows (|v| = 10 at H = 8192 reaches ~96 rows, OceanCurrents.cpp:294-299): the
// next dmg_ocean_advect gathers from a copy of v that is `reach` rows deep on each side that has a neighbour.  This call
// (re)allocates it if needed and copies the owned rows in; the host then refreshes its ghost rows like any other field's
// (dmg_ocean_halo with DMG_F_VADVECT, rows = reach) and calls dmg_ocean_advect.  Asynchronous on the context's stream.
int dmg_ocean_advect_prepare(dmg_ocean *o, int reach)
{
    int rc = require_terrain(o, "dmg_ocean_advect_prepare");
    if (rc) return rc;
    DMG_CHECK_ARG(reach >= 1);
    const int own = o->y1 - o->y0, W = o->grid.W;
    if (reach > o->grid.H / o->count)
        return dmg_fail(DMG_EINVAL, "advect reaches %d rows but a band owns only %d: a ghost slab must come from ONE neighbour "
                                    "(use fewer bands)", reach, o->grid.H / o->count);
    o->adv_rows = reach;
    o->adv_gs = o->rank > 0 ? reach : 0;
    o->adv_gn = o->rank < o->count - 1 ? reach : 0;
    const int cap_rows = own + 2 * ((reach + 31) & ~31);   // (grown in steps of 32 rows per side: the reach creeps up as currents develop)
    const size_t need = (size_t)cap_rows * W * sizeof(float2);
    if (need > o->vadv_cap) {
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));     // (the old copy may still be read by a queued advect)
        if (o->vadv) (void)hipFree(o->vadv);
        o->vadv = nullptr; o->vadv_cap = 0;
        if (hipMalloc((void **)&o->vadv, need) != hipSuccess)
            return dmg_fail(DMG_ENOMEM, "advect halo of %d rows: hipMalloc(%zu) failed", reach, need);
        o->vadv_cap = need;
    }
    DMG_HIP(hipMemcpyAsync(o->vadv + (size_t)o->adv_gs * W * 2, o->v[o->vcur] + (size_t)o->gs * W * 2, (size_t)own * W * sizeof(float2),
                           hipMemcpyDeviceToDevice, o->ctx->stream));
    o->adv_ready = true;
    return DMG_OK;
}

int dmg_ocean_advect(dmg_ocean *o)
{
    int rc = require_terrain(o, "dmg_ocean_advect");
    if (rc) return rc;
    pass_args a = make_args(o);
    const int stored = o->adv_ready ? o->adv_rows : o->ghost;
    if (o->adv_ready) {                                    // the deeper gather source of dmg_ocean_advect_prepare, this step only
        a.vg = (const float2 *)o->vadv; a.g_y_first = o->y0 - o->adv_gs; a.g_L = o->adv_gs + (o->y1 - o->y0) + o->adv_gn;
        o->adv_ready = false;
    }
    float k04 = powf(0.4f, 1.0f / 24.0f);                  // pow(vec2(0.4),vec2(1.0/24)) (:348), a constant
    if (o->count > 1) DMG_HIP(hipMemsetAsync(o->oob_flag, 0, sizeof(int), o->ctx->stream));
    rc = !(o->prm.quirks & DMG_QUIRK_FAST_TRANSCENDENTALS) ? launch_rows(o, k_advect<true>, a, o->gs, o->gs + (o->y1 - o->y0), k04)
                                                     : launch_rows(o, k_advect<false>, a, o->gs, o->gs + (o->y1 - o->y0), k04);
    if (rc) return rc;
    if (o->count > 1) {                                    // did a departure point leave the stored rows?
        int flag = 0;
        DMG_HIP(hipMemcpyAsync(&flag, o->oob_flag, sizeof(int), hipMemcpyDeviceToHost, o->ctx->stream));
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));
        if (flag)                                          // the result sits in the scratch buffer: v itself is untouched
            return dmg_fail(DMG_EINVAL, "advect: a departure point lies more than %d rows outside the band; create the "
                                        "ocean with ghost >= dmg_ocean_advect_reach(), or hand it deeper rows through "
                                        "dmg_ocean_advect_prepare", stored);
    }
    o->vcur ^= 1;                                          // v->swap(v_scratch) (:359)
    return DMG_OK;
}

// OceanCurrents::run body (:155-209)
int dmg_ocean_step(dmg_ocean *o, int n_outer)
{
    int rc = require_terrain(o, "dmg_ocean_step");
    if (rc) return rc;
    if (o->count != 1)
        return dmg_fail(DMG_EINVAL, "dmg_ocean_step is the single-band form; with bands the host interleaves the "
                                    "passes with halo refreshes (demiurge_amd/bands.py ocean_step)");
    DMG_CHECK_ARG(n_outer >= 0);
    for (int j = 0; j < n_outer; j++) {
        if ((rc = dmg_ocean_advect(o))) return rc;                                  // :158
        if ((rc = dmg_ocean_diffusion(o, o->prm.diffusion_iters))) return rc;       // :160
        if ((rc = dmg_ocean_divergence(o))) return rc;                              // :162
        if ((rc = dmg_ocean_jacobi(o, o->prm.jacobi_iters, 1))) return rc;          // :166
        if ((rc = dmg_ocean_subdiv(o))) return rc;                                  // :168
        o->n_outer++;
        // :170 divergence(p, divw_showcase) is a display-only field; :175-205 exports are dmg_ocean_get
    }
    return DMG_OK;
}

}  // extern "C"
