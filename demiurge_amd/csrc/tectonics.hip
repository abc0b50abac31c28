// Plate tectonics (reference src/filter/tectonics/Tectonics.cpp, Plate.cpp) on gfx950.
//
// The reference strings ~40 fragment-shader passes per step over RGBA32F textures: every plate carries its own
// texture (height, age) in its own rotating frame; each step the plates are "folded" onto a W x W global map through
// a rotated GL_NEAREST gather, gaps between plates are filled by a 10-pass ring search (oceanSpreading), a collision
// pass runs whose result nothing reads (see below), and every plate texture is rebuilt from the global map
// ("unfold") through the inverse rotation.
//
// Here a step touches the W x W map three times instead of ~40:
//   k_tect_fold    setzero + one fold per plate fused: the map texel only depends on itself, so the running value
//                  stays in registers across plates and the map is written once (16 B/texel instead of 3 x 32 B).
//                  Inside dmg_tectonics_step it also applies oceanSpreading's z := 0 and appends the texels no plate
//                  covers (x == 0) to an active list.
//   k_tect_ring_gather / k_tect_ring_scatter  x10: oceanSpreading's ring passes.  A texel that a plate owns has z = 0
//                  and can only be replaced by a candidate at distance < 0, i.e. never; so only the active (gap)
//                  texels are evaluated -- gather reads the unchanged map and writes new values beside it, scatter
//                  puts them back (the same Jacobi semantics as the reference's ping-pong), and the 10 passes cost
//                  O(gap texels) instead of 10 x 2 x 16 B x W^2.  The last scatter applies z := z>0.
//   k_tect_unfold  per plate: inverse-rotated gathers of the map; the 3x3 "is any neighbour mine" search stops at
//                  the first hit and is skipped where its outcome cannot change the texel
//   k_tect_render  once per dmg_tectonics_step call (the reference redraws the terrain every step for display)
//
// collision() (:399-618) writes texture c, which is read by the render pass only, and only its .y channel, which
// collision's last pass sets to plateIndices.x (:590).  Its other channels (which include reads of an uninitialised
// local, :459/:490, and 0/0, :494-495) never reach the terrain or the plates; they are not computed.  What collision
// does leave behind is the storage shuffle: c <-> plate swaps (:260) rotate three texture sizes (W x W, W x H, W x H)
// among c and the two plates with period 3, and the NEAREST resampling between those sizes IS observable.  It is
// reproduced exactly (c_w/c_h below).
//
// Arithmetic contract (bit-exact against oracle/tect_oracle.c, which is bit-exact against the reference GLSL on
// llvmpipe): IEEE binary32, no contraction; asin/atan/atan2 in the polynomial forms Mesa's GLSL compiler expands them
// to, sin/cos in the Cephes octant scheme with fused multiply-adds where gallivm emits them.  A gather is a texel
// CHOICE, so "close" transcendentals would move plate boundaries; these forms are the ones the reference's pixels
// were produced with.
#include "canvas.h"
#include "glsl_math.h"

namespace {

constexpr int TECT_MAX_PLATES = 8;

// ------------------------------------------------------------------------------------------------ transcendentals
// glsl_math.h: the forms Mesa / gallivm evaluate, bit for bit
__device__ __forceinline__ float t_asin(float x) { return glsl::asin(x); }
__device__ __forceinline__ float t_atan2(float y, float x) { return glsl::atan2(y, x); }
__device__ __forceinline__ void t_sincos(float x, float *s, float *c) { glsl::sincos(x, s, c); }
__device__ __forceinline__ float t_sin(float x) { return glsl::sin(x); }
__device__ __forceinline__ float t_cos(float x) { return glsl::cos(x); }

// ------------------------------------------------------------------------------------------------ geometry
struct ctex4 { const float4 *p; int w, h; };

__device__ __forceinline__ float4 t_fetch(const float4 *t, int w, int h, float u, float v)
{
    int i = min(max((int)floorf(u * (float)w), 0), w - 1);
    int j = min(max((int)floorf(v * (float)h), 0), h - 1);
    return t[(size_t)j * w + i];
}

// tex_to_spheric (Shader.h:49-53)
__device__ __forceinline__ void t_spheric(const canvas &c, float sx, float sy, float *lon, float *lat)
{
    *lon = sx * (c.c3 - c.c2) + c.c2;
    *lat = sy * (c.c1 - c.c0) + c.c0;
}

// offset(p, dp, resolution) (Shader.h:82-98) with an explicit resolution (the map is W x W, not the canvas size)
__device__ __forceinline__ void t_offset(const canvas &c, float px, float py, float dpx, float dpy, float rx, float ry,
                                         float *ox, float *oy)
{
    px = px + dpx / rx;
    py = py + dpy / ry;
    if (c.lonwrap) px = d_mod(px + 1.0f, 1.0f);
    if (c.spole && py < 0.0f) {
        py = -py;
        px = d_pole_lon(px * (c.c3 - c.c2) + c.c2);
        px = (px - c.c2) / (c.c3 - c.c2);
    }
    if (c.npole && py > 1.0f) {
        py = 2.0f - py;
        px = d_pole_lon(px * (c.c3 - c.c2) + c.c2);
        px = (px - c.c2) / (c.c3 - c.c2);
    }
    if (c.lonwrap) px = d_mod(px + 1.0f, 1.0f);
    *ox = px; *oy = py;
}

struct mat3 { float m[9]; };   // column-major, as glUniformMatrix3fv(.., GL_FALSE, ..) receives it

// plateTexture / inverseplateTexture (Tectonics.cpp:70-92): where st lands after rotating the sphere
__device__ __forceinline__ void t_rotate_st(const canvas &c, const mat3 &R, float lon, float lat, float *u, float *v)
{
    float sl, cl, so, co;
    t_sincos(lat, &sl, &cl);
    t_sincos(lon, &so, &co);
    float cx = cl * co, cy = cl * so, cz = sl;                                  // spheric_to_cartesian :61-63
    float rx = R.m[0] * cx + R.m[3] * cy + R.m[6] * cz;
    float ry = R.m[1] * cx + R.m[4] * cy + R.m[7] * cz;
    float rz = R.m[2] * cx + R.m[5] * cy + R.m[8] * cz;
    float lon2 = t_atan2(ry, rx), lat2 = t_asin(rz);                            // cartesian_to_spheric :65-67
    *u = (lon2 - c.c2) / (c.c3 - c.c2);                                         // spheric_to_tex :55-59
    *v = (lat2 - c.c0) / (c.c1 - c.c0);
}

// the same from the sines and cosines of lat and lon (the per-row / per-column tables below)
__device__ __forceinline__ void t_rotate_cs(const canvas &c, const mat3 &R, float sl, float cl, float so, float co, float *u, float *v)
{
    float cx = cl * co, cy = cl * so, cz = sl;
    float rx = R.m[0] * cx + R.m[3] * cy + R.m[6] * cz;
    float ry = R.m[1] * cx + R.m[4] * cy + R.m[7] * cz;
    float rz = R.m[2] * cx + R.m[5] * cy + R.m[8] * cz;
    float lon2 = t_atan2(ry, rx), lat2 = t_asin(rz);
    *u = (lon2 - c.c2) / (c.c3 - c.c2);
    *v = (lat2 - c.c0) / (c.c1 - c.c0);
}

// geodistance(p1, p2, size) (Shader.h:349-354); pow(x,2) is x*x
__device__ __forceinline__ float t_geodistance(const canvas &c, float p1x, float p1y, float p2x, float p2y, float sizex)
{
    float l1, a1, l2, a2;
    t_spheric(c, p1x, p1y, &l1, &a1);
    t_spheric(c, p2x, p2y, &l2, &a2);
    float s1 = t_sin(fabsf(a2 - a1) / 2.0f);
    float s2 = t_sin((l1 - l2) / 2.0f);
    float ds = 2.0f * t_asin(sqrtf(s1 * s1 + t_cos(a1) * t_cos(a2) * (s2 * s2)));
    return ds / (c.c3 - c.c2) * sizex;
}

// ------------------------------------------------------------------------------------------------ kernels
// plate initialisation, Tectonics.cpp:28-34 and :44-50: the terrain split at st.x = 0.5
__global__ void k_tect_init(canvas c, const float *img, float4 *p0, float4 *p1)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= c.W) return;
    float a = img[(size_t)y * c.W + x];
    float sx = ((float)x + 0.5f) / (float)c.W;
    float h0 = a > 0.0f ? a : -1.0f, h1 = a > 0.0f ? a : -2.0f;
    p0[(size_t)y * c.W + x] = sx > 0.5f ? make_float4(h0, 1.0f, 0.0f, 0.0f) : make_float4(0.0f, -1.0f, 0.0f, 0.0f);
    p1[(size_t)y * c.W + x] = sx < 0.5f ? make_float4(h1, 0.5f, 0.0f, 0.0f) : make_float4(0.0f, -1.0f, 0.0f, 0.0f);
}

struct fold_args {
    int n;
    ctex4 plate[TECT_MAX_PLATES];
    mat3 rot[TECT_MAX_PLATES];
};

// The rotated gathers of fold and unfold are done by TILES of TT_W x TT_H texels per workgroup (round 6).  A workgroup of 256
// texels of ONE row reads a one-texel-wide curve through the other frame: every 16-byte texel it fetches pulls in a cache line of which
// its row has no further use -- 9.8 GB moved for 2.15 GB of unavoidable traffic at 16384 x 8192 (profiles/r04_kernel_counters.txt: 4.6 x).
// A tile's image under a rotation is a compact patch: the lines it pulls in are used by its other rows.
#ifndef TT_W
#define TT_W 32
#define TT_H 8
#endif
static_assert(TT_W * TT_H == 256, "a tile is one workgroup of 256 threads");

// Round 6: fold and unfold are bound by their vector instructions (~380 per map texel: two sine / cosine pairs, then per plate a
// rotation, an arctangent, an arcsine, two IEEE divisions), not by their gathers -- tiling them took nothing off fold's 3.0 ms at
// 16384 x 8192.  The sine and cosine of a texel's longitude depend on its column alone and those of its latitude on its row:
// k_tect_trig evaluates them once per column and per row -- the same statements on the same operands, so the same bits -- and the
// kernels read two 8-byte entries instead of running ~120 instructions.  which 0: (sin, cos) of tex_to_spheric's lon for
// st.x = (i + 0.5) / n; 1: of its lat for st.y = (i + 0.5) / n.
__global__ void k_tect_trig(canvas c, int n, int which, float2 *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = ((float)i + 0.5f) / (float)n;
    // which 2: of the lon of the column offset(st, vec2(0)) hands back -- st.x through the two wraps of Shader.h:86,97, which
    // cost it its low bits (unfold's centre tap)
    if (which == 2 && c.lonwrap) { s = d_mod(s + 1.0f, 1.0f); s = d_mod(s + 1.0f, 1.0f); }
    float lon, lat, sn, cs;
    t_spheric(c, s, s, &lon, &lat);
    t_sincos(which == 1 ? lat : lon, &sn, &cs);
    out[i] = make_float2(sn, cs);
}

// Tectonics::fold (:278-293) = setzero (:99-106) then the fold shader (:112-149) once per plate, in registers
__global__ void __launch_bounds__(256) k_tect_fold(canvas c, fold_args A, float4 *map, int S, int zero_z,
                                                   unsigned *count, unsigned *active, const float2 *__restrict__ trig_lon,
                                                   const float2 *__restrict__ trig_lat)
{
    int x = blockIdx.x * TT_W + (int)(threadIdx.x % TT_W), y = blockIdx.y * TT_H + (int)(threadIdx.x / TT_W);
    if (x >= S || y >= S) return;
    const float2 tl = trig_lat[y], to = trig_lon[x];
    const float sl = tl.x, cl = tl.y, so = to.x, co = to.y;
    float cx = cl * co, cy = cl * so, cz = sl;
    float4 fc = make_float4(0.0f, -1.0f, -1.0f, -1.1e6f);          // plate id, height, age, collision
    for (int k = 0; k < A.n; k++) {
        const mat3 &R = A.rot[k];
        float rx = R.m[0] * cx + R.m[3] * cy + R.m[6] * cz;
        float ry = R.m[1] * cx + R.m[4] * cy + R.m[7] * cz;
        float rz = R.m[2] * cx + R.m[5] * cy + R.m[8] * cz;
        float u = (t_atan2(ry, rx) - c.c2) / (c.c3 - c.c2);
        float v = (t_asin(rz) - c.c0) / (c.c1 - c.c0);
        float4 p = t_fetch(A.plate[k].p, A.plate[k].w, A.plate[k].h, u, v);
        float index = (float)(k + 1);                              // index 0 is "no plate" (:283)
        float plateHeight = p.x, plateAge = p.y;
        float previousAge = fc.z, previousHeight = fc.y;
        bool overlap = plateAge >= 0.0f && previousAge >= 0.0f;
        if (overlap) fc.w = plateHeight;
        if (plateAge >= 0.0f && !overlap) fc = make_float4(index, plateHeight, plateAge, fc.w);
        bool landOnOcean = plateHeight > 0.0f && previousHeight <= 0.0f;
        bool plateYounger = plateAge < previousAge;
        bool plateOnTop = (plateYounger && plateHeight <= 0.0f && previousHeight <= 0.0f) ||
                          (!plateYounger && plateHeight > 0.0f && previousHeight > 0.0f) || landOnOcean;
        if (overlap && plateOnTop) fc = make_float4(index, plateHeight, plateAge, previousHeight);
    }
    if (fc.x == 0.0f) active[atomicAdd(count, 1u)] = (unsigned)y * (unsigned)S + (unsigned)x;
    if (zero_z) fc.z = 0.0f;                                       // oceanSpreading's first pass (:303-314)
    map[(size_t)y * S + x] = fc;
}

struct ring_table { float ct[16], sn[16]; };

// cos/sin of the 16 tap angles `2*3.14159*i/N` (:343), evaluated at run time like the shader's (not unrolled) loop does
__global__ void k_tect_ring_table(ring_table *t)
{
    int i = threadIdx.x;
    if (i < 16) t_sincos(2.0f * 3.14159f * (float)i / 16.0f, &t->sn[i], &t->ct[i]);
}

// oceanSpreading's first pass alone (:303-314), in place: only needed when dmg_tectonics_fold was called on its own
__global__ void k_tect_zero_z(float4 *map, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) map[i].z = 0.0f;
}

// oceanSpreading (:295-397), one ring pass (:325-350) over the active texels: newval[i] = shader(map, active[i])
__global__ void __launch_bounds__(256) k_tect_ring_gather(canvas c, const float4 *__restrict__ map, int S, float radius,
                                                          const ring_table *__restrict__ tab,
                                                          const unsigned *__restrict__ count,
                                                          const unsigned *__restrict__ active, float4 *__restrict__ newval)
{
    unsigned n = *count;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        unsigned idx = active[k];
        int x = (int)(idx % (unsigned)S), y = (int)(idx / (unsigned)S);
        float4 fc = map[idx];
        float sx = ((float)x + 0.5f) / (float)S, sy = ((float)y + 0.5f) / (float)S;
        float lon, phi;
        t_spheric(c, sx, sy, &lon, &phi);
        float factor = 1.0f / t_cos(fabsf(phi));
        for (int i = 0; i < 16; i++) {
            float nx, ny;
            t_offset(c, sx, sy, tab->ct[i] * radius * factor, tab->sn[i] * radius, (float)S, (float)S, &nx, &ny);
            float4 a = t_fetch(map, S, S, nx, ny);
            // the shader's `(nz < fc.z || fc.x == 0) && a.x != 0`: a tap that lands in the gap itself (a.x == 0) cannot
            // change the texel, so its distance -- most of a tap's ~250 instructions -- is not worked out; deep inside a
            // gap whole waves skip it
            if (a.x != 0.0f) {
                float nz = a.z + t_geodistance(c, nx, ny, sx, sy, (float)S);
                if (nz < fc.z || fc.x == 0.0f) fc = make_float4(a.x, -1.1f, nz, -1.1e6f);
            }
        }
        newval[k] = fc;
    }
}

// map[active[i]] = newval[i]; last: oceanSpreading's closing pass z := z>0 (:376-387); texels outside the list have z = 0
__global__ void k_tect_ring_scatter(float4 *__restrict__ map, const unsigned *__restrict__ count,
                                    const unsigned *__restrict__ active, const float4 *__restrict__ newval, int last)
{
    unsigned n = *count;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        float4 v = newval[k];
        if (last) v.z = v.z > 0.0f ? 1.0f : 0.0f;
        map[active[k]] = v;
    }
}

// back to plates (:216-251)
__global__ void __launch_bounds__(256) k_tect_unfold(canvas c, const float4 *__restrict__ map, int S, ctex4 old, mat3 inv,
                                                     float index, float4 *__restrict__ out, int w, int h,
                                                     const float2 *__restrict__ trig_lon, const float2 *__restrict__ trig_lat,
                                                     const float2 *__restrict__ trig_lon0)
{
    int x = blockIdx.x * TT_W + (int)(threadIdx.x % TT_W), y = blockIdx.y * TT_H + (int)(threadIdx.x / TT_W);
    if (x >= w || y >= h) return;
    float sx = ((float)x + 0.5f) / (float)w, sy = ((float)y + 0.5f) / (float)h;
    float u, v;
    const float2 tl = trig_lat[y], to = trig_lon[x];            // (sin, cos) of phi and lon: the row's and the column's
    t_rotate_cs(c, inv, tl.x, tl.y, to.x, to.y, &u, &v);
    float4 a = t_fetch(map, S, S, u, v);
    float4 fc = t_fetch(old.p, old.w, old.h, sx, sy);
    if (fc.y >= 0.0f) fc.y += 0.01f;                               // increment age
    // `differentIndex` (no tap of the 3x3 neighbourhood lands on this plate) only matters where it would change fc:
    // not where the land-keeps-its-place clause vetoes it, not where fc already is the "no crust" value.  Within the
    // search the first tap that is ours settles it; the centre tap goes first.
    const float4 none = make_float4(0.0f, -1.0f, 0.0f, 0.0f);
    bool is_none = __float_as_uint(fc.x) == 0u && fc.y == -1.0f && __float_as_uint(fc.z) == 0u && __float_as_uint(fc.w) == 0u;
    if (!(a.y <= 0.0f && fc.x > 0.0f) && !is_none) {
        // cos |phi| is the row's cosine (sincos takes |x| first; the sign only goes to the sine): the same bits from the table
        float factor = 1.0f / tl.y;
        // The centre tap is offset(st, vec2(0)): st.y itself and st.x through the two wraps -- a value of the column alone, so
        // its sine and cosine come from a table as well (k_tect_trig, which 2) and the tap is one rotation instead of
        // offset + two sincos + rotation.  (0 * factor is 0 for every finite factor; a row whose cosine is not a normal number
        // -- none on a sphere -- takes the loop from its first tap.)
        bool differentIndex = true;
        int k0 = 0;
        if (w == S && fabsf(factor) < 3.0e38f) {
            const float2 t0 = trig_lon0[x];
            t_rotate_cs(c, inv, tl.x, tl.y, t0.x, t0.y, &u, &v);
            differentIndex = t_fetch(map, S, S, u, v).x != index;
            k0 = 1;
        }
        for (int k = k0; k < 9 && differentIndex; k++) {
            int q = (k + 4) % 9, i = q / 3 - 1, j = q % 3 - 1;          // (0,0) first, then the reference's order
            float ox, oy, lo, la;
            t_offset(c, sx, sy, (float)i * factor, (float)j, (float)S, (float)S, &ox, &oy);
            t_spheric(c, ox, oy, &lo, &la);
            t_rotate_st(c, inv, lo, la, &u, &v);
            differentIndex = t_fetch(map, S, S, u, v).x != index;
        }
        if (differentIndex) fc = none;
    }
    if (fc.y < 0.0f && a.x == index && fabsf(a.z - 1.0f) < 0.01f) fc = make_float4(-index, 1.0f, 0.0f, 0.0f);
    out[(size_t)y * w + x] = fc;
}

// render (:186-210) of collision's c.y = plateIndices.x (:590): terrain = map.x, NEAREST
__global__ void k_tect_render(canvas c, const float4 *map, int S, float *terrain)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= c.W) return;
    float sx = ((float)x + 0.5f) / (float)c.W, sy = ((float)y + 0.5f) / (float)c.H;
    terrain[(size_t)y * c.W + x] = t_fetch(map, S, S, sx, sy).x;
}

// ------------------------------------------------------------------------------------------------ host side
// glm::rotate(mat4 m, float angle, vec3 axis) (GLM ext/matrix_transform.inl), m[col][row]
void glm_rotate(float m[4][4], float angle, const float v[3])
{
    float c = std::cos(angle), s = std::sin(angle);
    float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);     // glm::normalize
    float axis[3] = { v[0] * inv, v[1] * inv, v[2] * inv };
    float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0]; R[0][1] = temp[0] * axis[1] + s * axis[2]; R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2]; R[1][1] = c + temp[1] * axis[1]; R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1]; R[2][1] = temp[2] * axis[1] - s * axis[0]; R[2][2] = c + temp[2] * axis[2];
    float out[4][4];
    for (int k = 0; k < 3; k++)
        for (int r = 0; r < 4; r++) out[k][r] = m[0][r] * R[k][0] + m[1][r] * R[k][1] + m[2][r] * R[k][2];
    for (int r = 0; r < 4; r++) out[3][r] = m[3][r];
    memcpy(m, out, sizeof out);
}

}  // namespace

struct dmg_tectonics {
    dmg_ctx *ctx = nullptr;
    dmg_grid grid{};
    canvas cv{};
    int S = 0;                                   // the global map is S x S with S = W (Tectonics.cpp:22-24)
    int n = 0;                                   // plates
    float4 *map[2] = { nullptr, nullptr };       // a / b ping-pong; cur = the one holding the current global map
    int cur = 0;
    float4 *plate[TECT_MAX_PLATES] = {};
    int pw[TECT_MAX_PLATES] = {}, ph[TECT_MAX_PLATES] = {};
    float4 *cbuf = nullptr;                      // storage of the reference's texture c
    int c_w = 0, c_h = 0;
    float rotation[TECT_MAX_PLATES][4][4] = {};
    float av[TECT_MAX_PLATES][3] = {};           // Plate::angularVelocity
    float *terrain = nullptr;
    ring_table *ring = nullptr;
    unsigned *active = nullptr;                  // texels of the current map no plate covers, [0] of count = how many
    unsigned *count = nullptr;
    bool z_is_age = false;                       // map[cur] still carries the plate age in z (fold called on its own)
    long steps_done = 0;
    // (sin, cos) per column of a texture S wide, per row of one S high and of one H high (k_tect_trig): the storage shuffle
    // (c <-> plate) only ever makes textures of S x S and S x H
    float2 *trig_lon = nullptr, *trig_lat_S = nullptr, *trig_lat_H = nullptr, *trig_lon0 = nullptr;   // (trig_lon0: of the centre tap's column, k_tect_trig which 2)
    const float2 *trig_lat(int h) const { return h == S ? trig_lat_S : trig_lat_H; }
};

static mat3 plate_mat3(const dmg_tectonics *t, int k, bool transpose)
{
    mat3 r;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++)
            r.m[col * 3 + row] = transpose ? t->rotation[k][row][col] : t->rotation[k][col][row];
    return r;
}

static dim3 grid_for(int w, int h) { return dim3((unsigned)((w + 255) / 256), (unsigned)h); }
static dim3 grid_tiles(int w, int h) { return dim3((unsigned)((w + TT_W - 1) / TT_W), (unsigned)((h + TT_H - 1) / TT_H)); }

extern "C" {

int dmg_tectonics_create(dmg_ctx *ctx, const dmg_grid *grid, const float *terrain, dmg_tectonics **out)
{
    DMG_CHECK_ARG(ctx && grid && terrain && out);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_CHECK_ARG(grid->H <= grid->W);           // the reference's W x W scratch maps hold W x H plates
    DMG_CHECK_ARG((size_t)grid->W * grid->W <= ((size_t)1 << 31));
    dmg_tectonics *t = new dmg_tectonics;
    t->ctx = ctx;
    dmg_ctx_retain(ctx);
    t->grid = *grid;
    t->cv = make_canvas(*grid, topo);
    t->S = grid->W;
    t->n = 2;                                    // Tectonics.cpp:16-17
    size_t cap = (size_t)t->S * t->S * sizeof(float4);
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipMalloc((void **)&t->map[k], cap);
    for (int k = 0; k < t->n && e == hipSuccess; k++) e = hipMalloc((void **)&t->plate[k], cap);
    if (e == hipSuccess) e = hipMalloc((void **)&t->cbuf, cap);
    if (e == hipSuccess) e = hipMalloc((void **)&t->terrain, (size_t)grid->W * grid->H * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&t->ring, sizeof(ring_table));
    if (e == hipSuccess) e = hipMalloc((void **)&t->active, (size_t)t->S * t->S * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&t->trig_lon, (size_t)t->S * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void **)&t->trig_lat_S, (size_t)t->S * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void **)&t->trig_lat_H, (size_t)grid->H * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void **)&t->trig_lon0, (size_t)t->S * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void **)&t->count, sizeof(unsigned));
    if (e != hipSuccess) {
        dmg_tectonics_destroy(t);
        return dmg_fail(DMG_ENOMEM, "tectonics: hipMalloc failed: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < t->n; k++) {
        t->pw[k] = grid->W; t->ph[k] = grid->H;
        for (int i = 0; i < 4; i++) t->rotation[k][i][i] = 1.0f;                 // glm::mat4(1), Plate.cpp:13
    }
    t->c_w = t->S; t->c_h = t->S;
    // Tectonics.cpp:18,20: updateRotationBy(0.01, (-1,0,0)) and (-0.01, (-1,0,0))
    const float axis[3] = { -1.0f, 0.0f, 0.0f };
    for (int i = 0; i < 3; i++) { t->av[0][i] += 0.01f * axis[i]; t->av[1][i] += -0.01f * axis[i]; }
    e = hipMemcpyAsync(t->terrain, terrain, (size_t)grid->W * grid->H * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_tect_init, grid_for(grid->W, grid->H), dim3(256), 0, ctx->stream, t->cv, t->terrain,
                           t->plate[0], t->plate[1]);
        hipLaunchKernelGGL(k_tect_ring_table, dim3(1), dim3(64), 0, ctx->stream, t->ring);
        hipLaunchKernelGGL(k_tect_trig, dim3((unsigned)((t->S + 255) / 256)), dim3(256), 0, ctx->stream, t->cv, t->S, 0, t->trig_lon);
        hipLaunchKernelGGL(k_tect_trig, dim3((unsigned)((t->S + 255) / 256)), dim3(256), 0, ctx->stream, t->cv, t->S, 1, t->trig_lat_S);
        hipLaunchKernelGGL(k_tect_trig, dim3((unsigned)((grid->H + 255) / 256)), dim3(256), 0, ctx->stream, t->cv, grid->H, 1, t->trig_lat_H);
        hipLaunchKernelGGL(k_tect_trig, dim3((unsigned)((t->S + 255) / 256)), dim3(256), 0, ctx->stream, t->cv, t->S, 2, t->trig_lon0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);                   // `terrain` is the caller's buffer
    if (e != hipSuccess) {
        dmg_tectonics_destroy(t);
        return dmg_fail(DMG_EHIP, "tectonics: initialisation failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return DMG_OK;
}

void dmg_tectonics_destroy(dmg_tectonics *t)
{
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    for (int k = 0; k < 2; k++) if (t->map[k]) (void)hipFree(t->map[k]);
    for (int k = 0; k < TECT_MAX_PLATES; k++) if (t->plate[k]) (void)hipFree(t->plate[k]);
    if (t->cbuf) (void)hipFree(t->cbuf);
    if (t->terrain) (void)hipFree(t->terrain);
    if (t->ring) (void)hipFree(t->ring);
    if (t->active) (void)hipFree(t->active);
    if (t->count) (void)hipFree(t->count);
    for (void *q : { (void *)t->trig_lon, (void *)t->trig_lat_S, (void *)t->trig_lat_H, (void *)t->trig_lon0 }) if (q) (void)hipFree(q);
    dmg_ctx *ctx = t->ctx;
    delete t;
    dmg_ctx_release(ctx);
}

int dmg_tectonics_update_rotation(dmg_tectonics *t, int plate, float theta, const float axis[3])
{
    DMG_CHECK_ARG(t && axis && plate >= 0 && plate < t->n);
    for (int i = 0; i < 3; i++) t->av[plate][i] += theta * axis[i];              // Plate::updateRotationBy, Plate.cpp:46-48
    return DMG_OK;
}

static int tect_fold(dmg_tectonics *t, bool zero_z)
{
    fold_args A;
    A.n = t->n;
    for (int k = 0; k < t->n; k++) {
        float len = std::sqrt(t->av[k][0] * t->av[k][0] + t->av[k][1] * t->av[k][1] + t->av[k][2] * t->av[k][2]);
        DMG_CHECK_ARG(len > 0.0f);
        float inv = 1.0f / len;
        float unit[3] = { t->av[k][0] * inv, t->av[k][1] * inv, t->av[k][2] * inv };
        glm_rotate(t->rotation[k], len, unit);                                   // Plate::rotate, Plate.cpp:26-28
        A.plate[k] = ctex4{ t->plate[k], t->pw[k], t->ph[k] };
        A.rot[k] = plate_mat3(t, k, false);
    }
    DMG_HIP(hipMemsetAsync(t->count, 0, sizeof(unsigned), t->ctx->stream));
    hipLaunchKernelGGL(k_tect_fold, grid_tiles(t->S, t->S), dim3(256), 0, t->ctx->stream, t->cv, A, t->map[t->cur], t->S,
                       zero_z ? 1 : 0, t->count, t->active, t->trig_lon, t->trig_lat_S);
    DMG_HIP(hipGetLastError());
    t->z_is_age = !zero_z;
    return DMG_OK;
}

int dmg_tectonics_fold(dmg_tectonics *t)
{
    DMG_CHECK_ARG(t);
    return tect_fold(t, false);
}

int dmg_tectonics_spread(dmg_tectonics *t)
{
    DMG_CHECK_ARG(t);
    static const int exps[10] = { 0, 1, 2, 3, 4, 5, 4, 3, 2, 1 };                 // pow(2,i), :357-373
    hipStream_t st = t->ctx->stream;
    size_t n = (size_t)t->S * t->S;
    if (t->z_is_age) {
        hipLaunchKernelGGL(k_tect_zero_z, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t->map[t->cur], n);
        t->z_is_age = false;
    }
    // the list length lives on the device: a fixed grid walks it with a stride, no host round trip
    unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)t->ctx->num_cu * 16);
    float4 *newval = t->map[t->cur ^ 1];                                          // the idle half of the ping-pong pair
    for (int k = 0; k < 10; k++) {
        hipLaunchKernelGGL(k_tect_ring_gather, dim3(blocks), dim3(256), 0, st, t->cv, t->map[t->cur], t->S,
                           (float)(1 << exps[k]), t->ring, t->count, t->active, newval);
        hipLaunchKernelGGL(k_tect_ring_scatter, dim3(blocks), dim3(256), 0, st, t->map[t->cur], t->count, t->active,
                           newval, k == 9);
    }
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

int dmg_tectonics_unfold(dmg_tectonics *t)
{
    DMG_CHECK_ARG(t);
    for (int k = 0; k < t->n; k++) {
        std::swap(t->cbuf, t->plate[k]);                                         // c->swap(plate->getTexture()), :260
        std::swap(t->c_w, t->pw[k]);
        std::swap(t->c_h, t->ph[k]);
        hipLaunchKernelGGL(k_tect_unfold, grid_tiles(t->pw[k], t->ph[k]), dim3(256), 0, t->ctx->stream, t->cv,
                           t->map[t->cur], t->S, ctex4{ t->cbuf, t->c_w, t->c_h }, plate_mat3(t, k, true), (float)(k + 1),
                           t->plate[k], t->pw[k], t->ph[k], t->trig_lon, t->trig_lat(t->ph[k]), t->trig_lon0);
    }
    DMG_HIP(hipGetLastError());
    t->steps_done++;
    return DMG_OK;
}

int dmg_tectonics_render(dmg_tectonics *t)
{
    DMG_CHECK_ARG(t);
    hipLaunchKernelGGL(k_tect_render, grid_for(t->grid.W, t->grid.H), dim3(256), 0, t->ctx->stream, t->cv, t->map[t->cur],
                       t->S, t->terrain);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

int dmg_tectonics_step(dmg_tectonics *t, int n_steps)
{
    DMG_CHECK_ARG(t && n_steps >= 0);
    for (int i = 0; i < n_steps; i++) {
        int rc = tect_fold(t, true);
        if (!rc) rc = dmg_tectonics_spread(t);
        if (!rc && i == n_steps - 1) rc = dmg_tectonics_render(t);
        if (!rc) rc = dmg_tectonics_unfold(t);
        if (rc) return rc;
    }
    return DMG_OK;
}

int dmg_tectonics_plate_count(const dmg_tectonics *t) { return t ? t->n : 0; }

int dmg_tectonics_plate_size(const dmg_tectonics *t, int plate, int *w, int *h)
{
    DMG_CHECK_ARG(t && w && h && plate >= 0 && plate < t->n);
    *w = t->pw[plate]; *h = t->ph[plate];
    return DMG_OK;
}

int dmg_tectonics_get_plate(dmg_tectonics *t, int plate, float *rgba)
{
    DMG_CHECK_ARG(t && rgba && plate >= 0 && plate < t->n);
    DMG_HIP(hipMemcpyAsync(rgba, t->plate[plate], (size_t)t->pw[plate] * t->ph[plate] * sizeof(float4),
                           hipMemcpyDeviceToHost, t->ctx->stream));
    DMG_HIP(hipStreamSynchronize(t->ctx->stream));
    return DMG_OK;
}

int dmg_tectonics_get_map(dmg_tectonics *t, float *rgba)
{
    DMG_CHECK_ARG(t && rgba);
    DMG_HIP(hipMemcpyAsync(rgba, t->map[t->cur], (size_t)t->S * t->S * sizeof(float4), hipMemcpyDeviceToHost, t->ctx->stream));
    DMG_HIP(hipStreamSynchronize(t->ctx->stream));
    return DMG_OK;
}

extern "C++" const float *tectonics_terrain_device(const dmg_tectonics *t) { return t->terrain; }    // the render target (W x H), on the device

int dmg_tectonics_get_terrain(dmg_tectonics *t, float *terrain)
{
    DMG_CHECK_ARG(t && terrain);
    DMG_HIP(hipMemcpyAsync(terrain, t->terrain, (size_t)t->grid.W * t->grid.H * sizeof(float), hipMemcpyDeviceToHost,
                           t->ctx->stream));
    DMG_HIP(hipStreamSynchronize(t->ctx->stream));
    return DMG_OK;
}

}  // extern "C"
