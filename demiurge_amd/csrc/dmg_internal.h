// Internal declarations shared by the HIP translation units of libdemiurge_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/demiurge_hip.h"
#include "glsl_math.h"

// x / c for a constant c with r = RN(1/c):  q = x*r ; q' = fma(fma(-c, q, x), r, q)   (Markstein correction).
// Equals the IEEE quotient RN(x/c) for finite x with 2^-100 <= |x| <= 2^100 (and for x = 0); outside that range the
// remainder can underflow or the product overflow, and callers that can meet such values fall back to the IEEE
// sequence (dmg_div_rc_safe).  See oracle/dmg_oracle.c div_rc() and DESIGN.md section 4.
__device__ __forceinline__ float dmg_div_rc(float x, float c, float r)
{
    float q = __fmul_rn(x, r);
    return __fmaf_rn(__fmaf_rn(-c, q, x), r, q);
}
// true where dmg_div_rc(x, ...) is the IEEE quotient for certain (NaN and infinities compare false)
__device__ __forceinline__ bool dmg_div_rc_exact_for(float x)
{
    float a = fabsf(x);
    return x == 0.0f || (a >= 0x1p-100f && a <= 0x1p100f);
}

// Test hooks and experiment knobs (failure and delay injection, forced kernel forms, the unit cut's tuning constants, traces):
// values a test harness has handed over through dmg_debug_set(name, value) -- never the environment.  The library reads two
// environment variables of its own, both documented in include/demiurge_hip.h: DMG_MULTI_TIMEOUT_MS and DMG_ST_TWIN.
const char *dmg_hook(const char *name);              // nullptr: not set

// div2.hip: exact two-operation divisors, x / c[i] == fma(x, h[i], x * l[i]); ok[i] = 0 where no such pair exists
struct dmg_ctx;
int dmg_div2_splits(dmg_ctx *ctx, const float *c, int n, float *h, float *l, unsigned char *ok);

struct dmg_ctx {
    // scratch blocks handed back by finished calls, by size.  The filters that loop (erosion: ~45 temporaries per
    // flow solve, 1000 solves) would otherwise pay hipMalloc + a device-synchronising hipFree for each of them every
    // round.  One context = one stream = one caller thread at a time, so reuse is ordered by the stream itself.
    std::multimap<size_t, void *> scratch_pool;
    size_t scratch_pool_bytes = 0, scratch_cap = 0;
    std::mutex scratch_mu;          // the loopback transport of dmg_multi runs several band threads on ONE context: the pool's bookkeeping is theirs to share
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = 0;
    char name[128] = {0};
    // handles created on this context (dmg_ocean, dmg_tectonics, dmg_climate, dmg_erosion, dmg_multi) keep it alive:
    // dmg_destroy with live children only marks it, the last child's destroy releases it (dmg_ctx_release)
    // (the count starts at one: the owner's reference, given back by dmg_destroy)
    std::atomic<int> children{1};
    std::atomic<bool> destroy_requested{false};
    // the flow solve's memory between calls (flow.hip): how many rounds of pointer doubling the last solve of a grid this size needed -- the
    // next one asks whether anything still moved from that round on, not from the seventh (an erosion run's rivers change slowly, a
    // question is a synchronisation).  A hint: a wrong one costs a round or a question, never a result.
    // ... and the flow solve's per-row tables of the last grid (pixel sizes of the direction pass, cell areas of the sums: three floats a
    // row, the same every erosion round -- recomputed and uploaded per solve they were 2 048 cosines and three synchronisations of an
    // iteration); device memory of the context, released by dmg_trim
    // two streams of the flow solve's own and the events that fork and join them (created at first use): phase A of the upstream sums
    // runs beside the pass search and the host's growth, its small catchments beside its large ones
    hipStream_t flow_side = nullptr, flow_side2 = nullptr;
    hipEvent_t flow_fork = nullptr, flow_fork2 = nullptr, flow_join = nullptr, flow_join2 = nullptr, flow_copied = nullptr;
    dmg_grid flow_rows_grid = {};
    float *flow_rows = nullptr;         // [3][H]: dx, dy, area
    // ... and the blur's: cos|lat| per row of the last grid and room for the (0, r) passes' row records (blur_device: recomputed and
    // uploaded per call they were a host loop, a copy and a synchronisation of every erosion round); released by dmg_trim
    dmg_grid blur_rows_grid = {};
    void *blur_rows = nullptr;          // [H] floats, then BLUR_YREC_BYTES(H) of records (256-byte aligned)
    size_t flow_hint_cells = 0;
    int flow_hint_rounds = 0;
    unsigned flow_hint_age = 0;
    bool flow_hint_no_mouths = false;   // the last solve of flow_hint_cells cells found no river mouth: the next one looks for one first
};
void dmg_ctx_retain(dmg_ctx *ctx);
void dmg_ctx_release(dmg_ctx *ctx);

// scratch memory of a context (dmg_core.hip): exact-size reuse, released by dmg_trim / dmg_destroy
hipError_t dmg_scratch_alloc(dmg_ctx *ctx, size_t bytes, void **out);
void dmg_scratch_free(dmg_ctx *ctx, void *p, size_t bytes);

// thread-local error string behind dmg_last_error()
int dmg_fail(int code, const char *fmt, ...);

#define DMG_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return dmg_fail(DMG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                            \
    } while (0)

#define DMG_CHECK_ARG(cond)                                                                  \
    do {                                                                                     \
        if (!(cond)) return dmg_fail(DMG_EINVAL, "invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
    } while (0)

// ---- canvas geometry, the GLSL `cornerCoords` library (reference src/Shader.h:43-99) in fp32 ----
// GLSL's `#define M_PI 3.1415926535897932384626433832795` is a float literal.
static constexpr float DMG_PI = 3.1415926535897932384626433832795f;

struct dmg_topology {
    bool lonwrap;   // Shader.h:85  cornerCoords[2] < -M_PI+1e-4 && cornerCoords[3] > M_PI-1e-3
    bool spole;     // Shader.h:86  cornerCoords[0] < -M_PI/2+1e-4
    bool npole;     // Shader.h:91  cornerCoords[1] >  M_PI/2-1e-4
};

inline dmg_topology dmg_topology_of(const dmg_grid &g)
{
    dmg_topology t;
    t.lonwrap = (g.coords[2] < -DMG_PI + 1e-4f) && (g.coords[3] > DMG_PI - 1e-3f);
    t.spole = g.coords[0] < -DMG_PI / 2.0f + 1e-4f;
    t.npole = g.coords[1] > DMG_PI / 2.0f - 1e-4f;
    return t;
}

// tex_to_spheric(st).y for the texel row y (Shader.h:49-53), st.y = (y+0.5)/H
inline float dmg_row_lat(const dmg_grid &g, int y)
{
    float sy = ((float)y + 0.5f) / (float)g.H;
    return sy * (g.coords[1] - g.coords[0]) + g.coords[0];
}

// pixelsize(st) (Shader.h:71-74): vec2(dlon*cos(lat), dlat)*circumference/(2*M_PI)/textureSize
inline void dmg_row_pixelsize(const dmg_grid &g, int y, float *dx, float *dy)
{
    float lat = dmg_row_lat(g, y);
    float two_pi = 2.0f * DMG_PI;
    *dx = ((g.coords[3] - g.coords[2]) * glsl::cos(lat)) * g.circumference / two_pi / (float)g.W;   // the shader's runtime cos
    *dy = (g.coords[1] - g.coords[0]) * g.circumference / two_pi / (float)g.H;
}

// Validates the geometry this backend supports; fills topology.
int dmg_check_grid(const dmg_grid *g, dmg_topology *t);

// device-side hand-over between the tools of a coupled run (world.hip): no host round trips between the stages
struct dmg_erosion;
struct dmg_climate;
struct dmg_tectonics;
int erosion_create_from(dmg_ctx *ctx, const dmg_grid *grid, const float *h, bool on_device, int N, float exponent, float slope_exponent, float factor,
                        int lakes, dmg_erosion **out);
const float *erosion_heights_device(const dmg_erosion *e);
int climate_set_terrain_device(dmg_climate *k, const float *d_terrain);
int climate_set_terrain_peer(dmg_climate *k, const float *d_full, int src_device, hipEvent_t ready);
const float *tectonics_terrain_device(const dmg_tectonics *t);
