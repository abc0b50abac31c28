// Ocean currents: the reference's OceanCurrents filter (src/filter/OceanCurrents.cpp) as HIP kernels
// on row-major fp32 device buffers.  This file: object lifetime, field upload/download, latitude-band
// bookkeeping and the Jacobi pressure solve (the headline kernel).
//
// HBM layout of one band (rank r of n), all arrays row-major with W floats per row:
//     local row l  <->  grid row  y = y0 - gs + l,     l in [0, L),  L = gs + (y1 - y0) + gn
//   gs / gn = `ghost` rows kept towards the south / north neighbour band (0 at a grid edge).
//   p[2]   pressure ping-pong          (reference: `pressure` / `scratch`, OceanCurrents.cpp:44-45)
//   divw   right-hand side             (reference: `divw`, :54)
//   img    terrain heights, only > 0 matters (reference: Layer texture "img")
//   code   1 byte per cell, built once per terrain: which of the 4 neighbours / the cell itself is solid,
//          with the canvas topology (longitude wrap, pole reflection, CLAMP_TO_EDGE) already resolved
//   v      velocity, interleaved (vx,vy) like the reference's RG32F texture (:42-43)
//   rowc   per-local-row coefficients evaluated on the host in fp32 exactly as the GLSL does per fragment
#include "ocean_internal.h"

// ------------------------------------------------------------------------------------------------
// neighbour-code construction
// ------------------------------------------------------------------------------------------------
// bit0 N, bit1 S, bit2 E, bit3 W: "use the centre value instead of that neighbour"
//   = neighbour is land (OceanCurrents.cpp:470-473), or the tap is clamped onto the cell itself by
//     GL_CLAMP_TO_EDGE (src/Texture.cpp:29-30) on a non-wrapping edge -- both read pC.
// bit4 C: the cell itself is land -> result 0 (:481)
__global__ void k_build_codes(const float *__restrict__ img, unsigned char *__restrict__ code, int W, int L,
                              int y_first, int H, int lonwrap, int spole, int npole)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int l = blockIdx.y;
    if (x >= W) return;
    int y = y_first + l;
    const float *row = img + (size_t)l * W;
    unsigned c = 0;
    if (row[x] > 0.0f) c |= 16u;
    // east / west
    int xe = x + 1, xw = x - 1;
    if (xe >= W) { if (lonwrap) xe = 0; else xe = -1; }
    if (xw < 0) { if (lonwrap) xw = W - 1; else xw = -1; }
    if (xe < 0 || row[xe] > 0.0f) c |= 4u;
    if (xw < 0 || row[xw] > 0.0f) c |= 8u;
    // north
    if (y == H - 1) {
        if (npole) { if (row[(x + W / 2) % W] > 0.0f) c |= 1u; }     // offset(): y -> 2-y, lon += pi (Shader.h:91-95)
        else c |= 1u;
    } else if (l + 1 < L) {
        if (row[W + x] > 0.0f) c |= 1u;
    } else c |= 1u;                                                   // outermost ghost row: never computed
    // south
    if (y == 0) {
        if (spole) { if (row[(x + W / 2) % W] > 0.0f) c |= 2u; }     // Shader.h:86-90
        else c |= 2u;
    } else if (l >= 1) {
        if (row[x - W] > 0.0f) c |= 2u;
    } else c |= 2u;
    code[(size_t)l * W + x] = (unsigned char)c;
}

// p := 0 on land cells.  The reference never reads pressure of a land cell (a land neighbour is replaced
// by pC, a land centre yields 0), so this changes no result; it lets the sweeps treat land as exact zeros.
__global__ void k_zero_land(float *__restrict__ p, const unsigned char *__restrict__ code, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (code[i] & 16u)) p[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Jacobi sweep, one launch per sweep (general form: any W, any topology, bands)
// ------------------------------------------------------------------------------------------------
// Arithmetic contract (shared with ocean_jacobi_tb.hip and bit-exact with oracle/dmg_oracle.c
// orc_ocean_jacobi_rows): the GLSL statement  fc = ((pW + pE)/pw2.x + (pS + pN)/pw2.y - bC) / Beta
// (OceanCurrents.cpp:480) in its own order, each division by a row constant evaluated by dmg_div_rc().
__device__ __forceinline__ float jacobi_cell(float pC, float pN, float pS, float pE, float pW, float b, unsigned c,
                                             const jacobi_row &k)
{
    if (c & 1u) pN = pC;
    if (c & 2u) pS = pC;
    if (c & 4u) pE = pC;
    if (c & 8u) pW = pC;
    float r = jacobi_value(pN, pS, pE, pW, b, k);
    return (c & 16u) ? 0.0f : r;
}

struct sweep_args {
    const float *pin;
    float *pout;
    const float *b;
    const unsigned char *code;
    const float *cx, *cy, *beta, *rx, *ry, *rb;    // per local row: divisors and their reciprocals
    int W, L;
    int row_lo, row_hi;             // local rows computed by this sweep
    int l_south_pole, l_north_pole; // local row index of grid row 0 / H-1 when that edge is a pole here, else -1
    int lonwrap;
};

// 4 cells per thread, float4 loads/stores; requires W % 8 == 0 (so the pole shift W/2 keeps 16-byte alignment)
__global__ __launch_bounds__(256) void k_jacobi_sweep_v4(sweep_args a)
{
    const int W = a.W;
    int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int l = a.row_lo + blockIdx.y;
    if (x >= W) return;
    const float *rowC = a.pin + (size_t)l * W;
    // north / south source rows; at a pole the tap reflects into the same row, half a globe away
    const float *rowN, *rowS;
    int xn = x, xs = x;
    if (l == a.l_north_pole) { rowN = rowC; xn = x + W / 2; if (xn >= W) xn -= W; }
    else rowN = (l + 1 < a.L) ? rowC + W : rowC;
    if (l == a.l_south_pole) { rowS = rowC; xs = x + W / 2; if (xs >= W) xs -= W; }
    else rowS = (l >= 1) ? rowC - W : rowC;
    float4 pC = *reinterpret_cast<const float4 *>(rowC + x);
    float4 pN = *reinterpret_cast<const float4 *>(rowN + xn);
    float4 pS = *reinterpret_cast<const float4 *>(rowS + xs);
    int xw = x - 1, xe = x + 4;
    if (xw < 0) xw = a.lonwrap ? W - 1 : 0;
    if (xe >= W) xe = a.lonwrap ? 0 : W - 1;
    float pW = rowC[xw], pE = rowC[xe];
    float4 b = *reinterpret_cast<const float4 *>(a.b + (size_t)l * W + x);
    uchar4 c = *reinterpret_cast<const uchar4 *>(a.code + (size_t)l * W + x);
    jacobi_row k = { a.cx[l], a.cy[l], a.beta[l], a.rx[l], a.ry[l], a.rb[l] };
    float4 o;
    o.x = jacobi_cell(pC.x, pN.x, pS.x, pC.y, pW, b.x, c.x, k);
    o.y = jacobi_cell(pC.y, pN.y, pS.y, pC.z, pC.x, b.y, c.y, k);
    o.z = jacobi_cell(pC.z, pN.z, pS.z, pC.w, pC.y, b.z, c.z, k);
    o.w = jacobi_cell(pC.w, pN.w, pS.w, pE, pC.z, b.w, c.w, k);
    *reinterpret_cast<float4 *>(a.pout + (size_t)l * W + x) = o;
}

// one cell per thread, any W
__global__ __launch_bounds__(256) void k_jacobi_sweep_s(sweep_args a)
{
    const int W = a.W;
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int l = a.row_lo + blockIdx.y;
    if (x >= W) return;
    const float *rowC = a.pin + (size_t)l * W;
    const float *rowN, *rowS;
    int xn = x, xs = x;
    if (l == a.l_north_pole) { rowN = rowC; xn = (x + W / 2) % W; }
    else rowN = (l + 1 < a.L) ? rowC + W : rowC;
    if (l == a.l_south_pole) { rowS = rowC; xs = (x + W / 2) % W; }
    else rowS = (l >= 1) ? rowC - W : rowC;
    int xw = x - 1, xe = x + 1;
    if (xw < 0) xw = a.lonwrap ? W - 1 : 0;
    if (xe >= W) xe = a.lonwrap ? 0 : W - 1;
    size_t i = (size_t)l * W + x;
    jacobi_row k = { a.cx[l], a.cy[l], a.beta[l], a.rx[l], a.ry[l], a.rb[l] };
    a.pout[i] = jacobi_cell(rowC[x], rowN[xn], rowS[xs], rowC[xe], rowC[xw], a.b[i], a.code[i], k);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int launch_sweep(dmg_ocean *o, int row_lo, int row_hi)
{
    if (row_hi <= row_lo) return DMG_OK;
    sweep_args a;
    a.pin = o->p[o->cur];
    a.pout = o->p[o->cur ^ 1];
    a.b = o->divw;
    a.code = o->code;
    a.cx = o->jac_cx; a.cy = o->jac_cy; a.beta = o->jac_beta;
    a.rx = o->jac_rx; a.ry = o->jac_ry; a.rb = o->jac_rb;
    a.W = o->grid.W; a.L = o->L;
    a.row_lo = row_lo; a.row_hi = row_hi;
    a.l_south_pole = (o->topo.spole && o->y0 == 0) ? o->gs : -1;
    a.l_north_pole = (o->topo.npole && o->y1 == o->grid.H) ? o->gs + (o->y1 - o->y0) - 1 : -1;
    a.lonwrap = o->topo.lonwrap ? 1 : 0;
    const int W = o->grid.W;
    o->last_kernel = W % 8 == 0 ? "k_jacobi_sweep_v4" : "k_jacobi_sweep_s";
    if (W % 8 == 0) {
        dim3 grid((W / 4 + 255) / 256, row_hi - row_lo);
        hipLaunchKernelGGL(k_jacobi_sweep_v4, grid, dim3(256), 0, o->ctx->stream, a);
    } else {
        dim3 grid((W + 255) / 256, row_hi - row_lo);
        hipLaunchKernelGGL(k_jacobi_sweep_s, grid, dim3(256), 0, o->ctx->stream, a);
    }
    DMG_HIP(hipGetLastError());
    o->cur ^= 1;
    return DMG_OK;
}

int ocean_build_codes(dmg_ocean *o)
{
    dim3 grid((o->grid.W + 255) / 256, o->L);
    hipLaunchKernelGGL(k_build_codes, grid, dim3(256), 0, o->ctx->stream, o->img, o->code, o->grid.W, o->L,
                       o->y0 - o->gs, o->grid.H, (int)o->topo.lonwrap, (int)o->topo.spole, (int)o->topo.npole);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

static int zero_land(dmg_ocean *o, float *p)
{
    size_t n = (size_t)o->L * o->grid.W;
    hipLaunchKernelGGL(k_zero_land, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, o->ctx->stream, p, o->code, n);
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

int ocean_zero_land(dmg_ocean *o)
{
    int rc = zero_land(o, o->p[0]);
    return rc ? rc : zero_land(o, o->p[1]);
}

// fills a local (band + ghosts) device array from a FULL-grid host array
int ocean_upload_rows(dmg_ocean *o, float *dst, const float *src_full, int comps)
{
    size_t row = (size_t)o->grid.W * comps * sizeof(float);
    const float *src = src_full + (size_t)(o->y0 - o->gs) * o->grid.W * comps;
    DMG_HIP(hipMemcpyAsync(dst, src, row * o->L, hipMemcpyHostToDevice, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));   // the caller's buffer may be freed after return
    return DMG_OK;
}

// copies the OWNED rows of a local device array into a FULL-grid host array
int ocean_download_rows(dmg_ocean *o, float *dst_full, const float *src, int comps)
{
    size_t row = (size_t)o->grid.W * comps * sizeof(float);
    DMG_HIP(hipMemcpyAsync(dst_full + (size_t)o->y0 * o->grid.W * comps, src + (size_t)o->gs * o->grid.W * comps,
                           row * (o->y1 - o->y0), hipMemcpyDeviceToHost, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    return DMG_OK;
}

extern "C" {

void dmg_ocean_params_default(dmg_ocean_params *p)
{
    if (!p) return;
    p->pressurefactor = 100.0f;   // OceanCurrents.h:21
    p->jacobi_iters = 5000;       // OceanCurrents.cpp:50
    p->diffusion_iters = 50;      // :701
    p->timestep_h = 24.0f;        // :294
    p->scale = 420.0f;            // :419,:478,:574,:689
    p->wind = 10.0f;              // :342
    p->quirks = DMG_QUIRK_DIFFUSION_SCALAR_RHS;   // :687,:693; the reference's pixels bit for bit (no tolerance bit set)
    p->jacobi_kernel = DMG_JACOBI_AUTO;
}

int dmg_ocean_create(dmg_ctx *ctx, const dmg_grid *grid, const dmg_ocean_params *params, int band_rank,
                     int band_count, int ghost, dmg_ocean **out)
{
    DMG_CHECK_ARG(ctx != nullptr && out != nullptr);
    *out = nullptr;
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_CHECK_ARG(band_count >= 1 && band_rank >= 0 && band_rank < band_count);
    DMG_CHECK_ARG(ghost >= 1);
    DMG_CHECK_ARG(grid->H / band_count >= ghost);     // a ghost slab must come from ONE neighbour band
    DMG_HIP(hipSetDevice(ctx->device));
    dmg_ocean *o = new (std::nothrow) dmg_ocean();
    if (!o) return dmg_fail(DMG_ENOMEM, "out of host memory");
    o->ctx = ctx;
    dmg_ctx_retain(ctx);
    o->grid = *grid;
    o->topo = topo;
    if (params) o->prm = *params; else dmg_ocean_params_default(&o->prm);
    o->rank = band_rank; o->count = band_count; o->ghost = ghost;
    o->y0 = dmg_band_row0(grid->H, band_count, band_rank);
    o->y1 = dmg_band_row0(grid->H, band_count, band_rank + 1);
    o->gs = band_rank > 0 ? ghost : 0;
    o->gn = band_rank < band_count - 1 ? ghost : 0;
    o->L = o->gs + (o->y1 - o->y0) + o->gn;
    size_t n = (size_t)o->L * grid->W;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc((void **)&o->p[0], n * 4); alloc((void **)&o->p[1], n * 4);
    alloc((void **)&o->divw, n * 4); alloc((void **)&o->img, n * 4);
    alloc((void **)&o->code, n);
    alloc((void **)&o->code8, n);
    alloc((void **)&o->v[0], n * 8); alloc((void **)&o->v[1], n * 8);
    alloc((void **)&o->rowtab, sizeof(float) * ROWTAB_FIELDS * o->L);
    alloc((void **)&o->jcoef, sizeof(float) * 8 * o->L);
    alloc((void **)&o->jcoefp, sizeof(float) * 4 * (o->L + 32));
    alloc((void **)&o->jcoef2, sizeof(float) * 8 * o->L);
    alloc((void **)&o->jcoef3, sizeof(float) * 8 * o->L);
    alloc((void **)&o->rowbad, (size_t)o->L);
    alloc((void **)&o->oob_flag, sizeof(int));
    if (e != hipSuccess) {
        dmg_ocean_destroy(o);
        return dmg_fail(DMG_ENOMEM, "hipMalloc of %zu cells failed: %s", n, hipGetErrorString(e));
    }
    rc = ocean_upload_row_tables(o);
    if (rc) { dmg_ocean_destroy(o); return rc; }
    hipStream_t s = ctx->stream;
    // OceanCurrents::OceanCurrents: v := 0 (:92,:102); pressure is zeroed at every solve (:495-497)
    if (hipMemsetAsync(o->p[0], 0, n * 4, s) != hipSuccess || hipMemsetAsync(o->p[1], 0, n * 4, s) != hipSuccess ||
        hipMemsetAsync(o->divw, 0, n * 4, s) != hipSuccess || hipMemsetAsync(o->v[0], 0, n * 8, s) != hipSuccess ||
        hipMemsetAsync(o->v[1], 0, n * 8, s) != hipSuccess) {
        dmg_ocean_destroy(o);
        return dmg_fail(DMG_EHIP, "hipMemsetAsync failed");
    }
    *out = o;
    return DMG_OK;
}

void dmg_ocean_destroy(dmg_ocean *o)
{
    if (!o) return;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    for (void *p : { (void *)o->p[0], (void *)o->p[1], (void *)o->divw, (void *)o->img, (void *)o->code, (void *)o->code8,
                     (void *)o->v[0], (void *)o->v[1], (void *)o->rowtab, (void *)o->jcoef, (void *)o->jcoefp, (void *)o->jcoef2, (void *)o->jcoef3, (void *)o->rowbad, (void *)o->oob_flag, (void *)o->rowclass, (void *)o->vadv, (void *)o->deep_err })
        if (p) (void)hipFree(p);
    ocean_tb_free_lists(o);
    if (o->ev_fork) (void)hipEventDestroy(o->ev_fork);
    if (o->ev_edge) (void)hipEventDestroy(o->ev_edge);
    dmg_ctx *ctx = o->ctx;
    delete o;
    dmg_ctx_release(ctx);
}

int dmg_ocean_set_terrain(dmg_ocean *o, const float *img)
{
    DMG_CHECK_ARG(o != nullptr && img != nullptr);
    int rc = ocean_upload_rows(o, o->img, img, 1);
    if (rc) return rc;
    return ocean_terrain_changed(o);
}

// everything derived from o->img: neighbour codes, the blocked kernel's row classes, and p := +0 on land
extern "C++" int ocean_terrain_changed(dmg_ocean *o)
{
    int rc = ocean_build_codes(o);
    if (rc) return rc;
    rc = ocean_build_codes8(o);
    if (rc) return rc;
    rc = ocean_build_rowclass(o);
    if (rc) return rc;
    o->have_terrain = true;
    // keep the "land holds exact +0 in BOTH ping-pong buffers" invariant (the blocked kernel neither loads nor
    // stores wave-rows that are land everywhere)
    if ((rc = zero_land(o, o->p[0]))) return rc;
    return zero_land(o, o->p[1]);
}

int dmg_ocean_set_divw(dmg_ocean *o, const float *divw)
{
    DMG_CHECK_ARG(o != nullptr && divw != nullptr);
    return ocean_upload_rows(o, o->divw, divw, 1);
}

int dmg_ocean_set_pressure(dmg_ocean *o, const float *pressure)
{
    DMG_CHECK_ARG(o != nullptr && pressure != nullptr);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_set_terrain must come first");
    int rc = no_pending(o, "dmg_ocean_set_pressure");
    if (rc) return rc;
    rc = ocean_upload_rows(o, o->p[o->cur], pressure, 1);
    if (rc) return rc;
    return zero_land(o, o->p[o->cur]);      // the other buffer already holds 0 on land
}

int dmg_ocean_zero_pressure(dmg_ocean *o)
{
    DMG_CHECK_ARG(o != nullptr);
    { int prc = no_pending(o, "dmg_ocean_zero_pressure"); if (prc) return prc; }
    // OceanCurrents.cpp:495-497  p->apply(setzero, pressure)
    DMG_HIP(hipMemsetAsync(o->p[o->cur], 0, (size_t)o->L * o->grid.W * 4, o->ctx->stream));
    return DMG_OK;
}

int dmg_ocean_jacobi_sweeps(dmg_ocean *o, int n)
{
    DMG_CHECK_ARG(o != nullptr && n >= 0);
    { int prc = no_pending(o, "dmg_ocean_jacobi_sweeps"); if (prc) return prc; }
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_set_terrain must come first");
    if (o->count > 1 && n > o->ghost)
        return dmg_fail(DMG_EINVAL, "%d sweeps between halo refreshes but only %d ghost rows", n, o->ghost);
    DMG_HIP(hipSetDevice(o->ctx->device));
    if (o->prm.quirks & DMG_QUIRK_LLVMPIPE_RASTER) {     // opt-in: the software GL path's pixels on any grid size
        for (int k = 0; k < n; k++) {
            int rc = ocean_launch_literal(o);
            if (rc) return rc;
            o->cur ^= 1;
            o->n_launches++; o->n_sweeps++;
        }
        return DMG_OK;
    }
    bool blocked = o->prm.jacobi_kernel != DMG_JACOBI_SINGLE && ocean_tb_supported(o);
    if (o->prm.jacobi_kernel == DMG_JACOBI_BLOCKED && !blocked)
        return dmg_fail(DMG_EINVAL, "DMG_JACOBI_BLOCKED needs a full-longitude grid with W %% 8 == 0 and W >= 256");
    // after k sweeps since the last halo refresh the rows still computable lie k rows inside the ghost slabs:
    // the region shrinks by one row per sweep on every side that has a neighbour band
    if (blocked) {
        static const int cap = [] { const char *e = dmg_hook("DMG_TB_SWEEPS"); int v = e ? atoi(e) : 8; return v >= 1 && v <= 8 ? v : 8; }();
        // the deep form (16 sweeps per trip through HBM) where it pays: the tolerance rendering (ocean_jacobi_deep.hip)
        const bool deep = cap == 8 && ocean_deep_enabled(o);
        for (int k = 0; k < n;) {
            if (deep && n - k >= 16) {
                k += 16;
                int lo = o->gs ? k : 0, hi = o->L - (o->gn ? k : 0);
                int rc = ocean_launch_deep(o, lo, hi, o->cur, o->ctx->stream);
                if (rc) return rc;
                o->cur ^= 1;
                o->n_launches++; o->n_sweeps += 16;
                continue;
            }
            int m = n - k < cap ? n - k : cap;   // DMG_TB_SWEEPS (1..8): sweeps per launch, for phase-cost experiments
            k += m;
            int lo = o->gs ? k : 0, hi = o->L - (o->gn ? k : 0);
            int rc = ocean_launch_tb(o, lo, hi, m);
            if (rc) return rc;
            o->cur ^= 1;
            o->n_launches++; o->n_sweeps += m;
        }
        return DMG_OK;                                   // (the deep form's error word is looked at where results leave the device: dmg_ocean_get)
    }
    for (int k = 1; k <= n; k++) {
        int lo = o->gs ? k : 0, hi = o->L - (o->gn ? k : 0);
        int rc = launch_sweep(o, lo, hi);
        if (rc) return rc;
        o->n_launches++; o->n_sweeps++;
    }
    return DMG_OK;
}

// A block of sweeps split so that the halo exchange of the NEXT block overlaps this block's last launch:
//   _begin  runs every launch of the block but the last, then only those tiles of the last launch that produce the rows
//           the neighbour bands wait for (the outermost `ghost` owned rows on each side), and makes that buffer current
//           -- dmg_ocean_halo() now returns slabs of the new iterate, the host starts its sends and receives;
//   _end    sweeps the remaining (interior) tiles of the last launch, which neither read the ghost rows being received
//           nor write the rows being sent.
// Both parts belong to one tiling of the same launch: results are bit-identical to dmg_ocean_jacobi_sweeps(o, n).
// Without the blocked kernel (or with a single band) _begin does the whole block and _end nothing.
int dmg_ocean_jacobi_sweeps_begin(dmg_ocean *o, int n) { return ocean_jacobi_sweeps_begin(o, n, nullptr); }

// edge_stream: the stream that carries the edge units of the block's last launch (null: the context's).  On a stream of
// its own the edge part starts together with the interior part (dmg_ocean_jacobi_sweeps_end, on the context's stream) as soon
// as the block's launch before it is done -- both read the previous iterate and write disjoint rows of the new one -- so the
// split costs no chip time, and whatever the caller queues behind the edge part on that stream (the halo exchange) runs
// while the interior is still being swept.
extern "C++" int ocean_jacobi_sweeps_begin(dmg_ocean *o, int n, hipStream_t edge_stream)
{
    DMG_CHECK_ARG(o != nullptr && n >= 0);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_set_terrain must come first");
    int rc = no_pending(o, "dmg_ocean_jacobi_sweeps_begin");
    if (rc) return rc;
    const bool blocked = o->prm.jacobi_kernel != DMG_JACOBI_SINGLE && ocean_tb_supported(o);
    static const int cap = [] { const char *e = dmg_hook("DMG_TB_SWEEPS"); int v = e ? atoi(e) : 8; return v >= 1 && v <= 8 ? v : 8; }();
    if (!blocked || o->count == 1 || n == 0 || cap != 8) return dmg_ocean_jacobi_sweeps(o, n);
    if (n > o->ghost)
        return dmg_fail(DMG_EINVAL, "%d sweeps between halo refreshes but only %d ghost rows", n, o->ghost);
    DMG_HIP(hipSetDevice(o->ctx->device));
    const int last = n - ((n - 1) / 8) * 8;              // sweeps of the block's last launch (1..8)
    if (n > last) { rc = dmg_ocean_jacobi_sweeps(o, n - last); if (rc) return rc; }
    const int k = n;                                     // sweeps since the last halo refresh once this launch is done
    const int lo = o->gs ? k : 0, hi = o->L - (o->gn ? k : 0);
    const int zone_lo = o->gs ? std::min(o->gs + o->ghost, hi) : lo;      // end of the rows sent southwards
    const int zone_hi = o->gn ? std::max(o->L - o->gn - o->ghost, lo) : hi;   // start of the rows sent northwards
    // while the tile-shape tuner of this launch geometry is still timing its candidates (the first six launches of a
    // geometry below 4096 tiles), issue the launch whole so that it gets timed: the split starts once a shape is chosen
    if (last == 8 && ocean_tb_tuning_open(o, lo, hi)) {
        rc = ocean_launch_tb(o, lo, hi, last);
        if (rc) return rc;
        o->cur ^= 1;
        o->n_launches++; o->n_sweeps += last;
        return DMG_OK;
    }
    if (edge_stream && edge_stream != o->ctx->stream) {
        // (kernel -> kernel on one device: an agent-scope release is all the edge units need)
        static const int evf = [] { const char *e = dmg_hook("DMG_MULTI_EVENT_FLAGS"); return e ? atoi(e) : 1; }();
        if (!o->ev_fork) DMG_HIP(hipEventCreateWithFlags(&o->ev_fork, hipEventDisableTiming | (evf >= 1 ? hipEventReleaseToDevice : 0)));
        DMG_HIP(hipEventRecord(o->ev_fork, o->ctx->stream));             // everything this launch reads has been written
        DMG_HIP(hipStreamWaitEvent(edge_stream, o->ev_fork, 0));
    }
    rc = ocean_launch_tb(o, lo, hi, last, 1, zone_lo, zone_hi, -1, edge_stream);
    if (rc) return rc;
    // Edge first: the interior units start when the edge units are done, not beside them.  Beside a full launch the edge
    // units -- a tenth of the launch's units -- take as long as the launch itself (47-74 us on a 1024-row band: they share
    // every SIMD with an interior wave), so the exchange behind them started when the interior was nearly through and
    // hid nothing (profiles/r03_fake_link.log).  Alone on the chip, as units of 8 rows (st_cut's edge_seg), they are done
    // in ~25 us, and the exchange then has the whole interior to hide behind.  DMG_MULTI_EDGE_FIRST=0: side by side.
    static const bool edge_first = [] { const char *e = dmg_hook("DMG_MULTI_EDGE_FIRST"); return !e || atoi(e) != 0; }();
    o->edge_wait = false;
    if (edge_first && edge_stream && edge_stream != o->ctx->stream) {
        if (!o->ev_edge) DMG_HIP(hipEventCreateWithFlags(&o->ev_edge, hipEventDisableTiming));
        DMG_HIP(hipEventRecord(o->ev_edge, edge_stream));
        o->edge_wait = true;
    }
    o->pending.active = true;
    o->pending.lo = lo; o->pending.hi = hi; o->pending.n = last; o->pending.zone_lo = zone_lo; o->pending.zone_hi = zone_hi;
    o->cur ^= 1;                                         // the slabs handed out from now on are the new iterate's
    o->n_launches++; o->n_sweeps += last;
    return DMG_OK;
}

int dmg_ocean_jacobi_sweeps_end(dmg_ocean *o)
{
    DMG_CHECK_ARG(o != nullptr);
    if (!o->pending.active) return DMG_OK;
    DMG_HIP(hipSetDevice(o->ctx->device));
    o->cur ^= 1;                                         // read the previous iterate, write the one made current by _begin
    if (o->edge_wait) { DMG_HIP(hipStreamWaitEvent(o->ctx->stream, o->ev_edge, 0)); o->edge_wait = false; }
    int rc = ocean_launch_tb(o, o->pending.lo, o->pending.hi, o->pending.n, 2, o->pending.zone_lo, o->pending.zone_hi);
    o->cur ^= 1;
    o->pending.active = false;
    return rc;
}

// ---- pieces for a host that schedules the launches of a block itself (demiurge_amd/bands.py, pipelined solve) ----
// rows [lo, hi) the 8-sweep launch computes that ends `k` sweeps after the last halo refresh, and the output rows per
// tile row of the shape it runs with (0 while that geometry's shape tuner is still open, or without the blocked kernel)
int dmg_ocean_jacobi_geometry(dmg_ocean *o, int k, int *lo, int *hi, int *tile_rows)
{
    DMG_CHECK_ARG(o != nullptr && lo != nullptr && hi != nullptr && tile_rows != nullptr && k >= 0);
    *lo = o->gs ? k : 0;
    *hi = o->L - (o->gn ? k : 0);
    const bool blocked = o->have_terrain && o->prm.jacobi_kernel != DMG_JACOBI_SINGLE && ocean_tb_supported(o);
    *tile_rows = blocked ? ocean_tb_tile_rows(o, *lo, *hi) : 0;
    return DMG_OK;
}

int dmg_ocean_band_rows(dmg_ocean *o, int *stored_rows, int *ghost_south, int *ghost_north)
{
    DMG_CHECK_ARG(o != nullptr);
    if (stored_rows) *stored_rows = o->L;
    if (ghost_south) *ghost_south = o->gs;
    if (ghost_north) *ghost_north = o->gn;
    return DMG_OK;
}

// n (1..8) sweeps over local rows [lo, hi) from pressure buffer `src` (0/1) into the other one, on `stream` (a
// hipStream_t; NULL = the context's).  part 0 = every tile row, 1 = the tile rows that produce rows below zone_lo or
// above zone_hi, 2 = the others (one tiling, anchored at lo).  Changes neither the current buffer nor the counters.
int dmg_ocean_jacobi_launch(dmg_ocean *o, int lo, int hi, int n, int part, int zone_lo, int zone_hi, int src, void *stream)
{
    DMG_CHECK_ARG(o != nullptr && n >= 1 && n <= 8 && part >= 0 && part <= 2 && (src == 0 || src == 1));
    DMG_CHECK_ARG(lo >= 0 && hi <= o->L);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_set_terrain must come first");
    if (o->prm.jacobi_kernel == DMG_JACOBI_SINGLE || !ocean_tb_supported(o))
        return dmg_fail(DMG_EINVAL, "dmg_ocean_jacobi_launch needs the blocked kernel");
    int rc = no_pending(o, "dmg_ocean_jacobi_launch");
    if (rc) return rc;
    DMG_HIP(hipSetDevice(o->ctx->device));
    return ocean_launch_tb(o, lo, hi, n, part, zone_lo, zone_hi, src, (hipStream_t)stream);
}

int dmg_ocean_pressure_buffer(dmg_ocean *o, int *current)
{
    DMG_CHECK_ARG(o != nullptr && current != nullptr);
    *current = o->cur;
    return DMG_OK;
}

// after a self-scheduled block: buffer `idx` holds the new iterate; account for `launches` launches of `sweeps` sweeps
int dmg_ocean_set_pressure_buffer(dmg_ocean *o, int idx, int launches, int sweeps)
{
    DMG_CHECK_ARG(o != nullptr && (idx == 0 || idx == 1) && launches >= 0 && sweeps >= 0);
    int rc = no_pending(o, "dmg_ocean_set_pressure_buffer");
    if (rc) return rc;
    o->cur = idx;
    o->n_launches += launches; o->n_sweeps += sweeps;
    return DMG_OK;
}

// the pressure slabs of dmg_ocean_halo, but of buffer `buffer` (0/1) instead of the current one
int dmg_ocean_pressure_halo(dmg_ocean *o, int buffer, int side, int rows, void **send, void **recv, size_t *bytes)
{
    DMG_CHECK_ARG(o != nullptr && (buffer == 0 || buffer == 1));
    const int keep = o->cur;
    o->cur = buffer;
    int rc = dmg_ocean_halo(o, DMG_F_PRESSURE, side, rows, send, recv, bytes);
    o->cur = keep;
    return rc;
}

int dmg_ocean_counters(dmg_ocean *o, long long *jacobi_launches, long long *jacobi_sweeps)
{
    DMG_CHECK_ARG(o != nullptr);
    if (jacobi_launches) *jacobi_launches = o->n_launches;
    if (jacobi_sweeps) *jacobi_sweeps = o->n_sweeps;
    return DMG_OK;
}

int dmg_ocean_jacobi_kernel_name(dmg_ocean *o, char *buf, size_t len)
{
    DMG_CHECK_ARG(o != nullptr && buf != nullptr && len > 0);
    snprintf(buf, len, "%s", o->last_kernel);
    return DMG_OK;
}

int dmg_ocean_jacobi(dmg_ocean *o, int iters, int zero_first)
{
    DMG_CHECK_ARG(o != nullptr && iters >= 0);
    if (o->count != 1) return dmg_fail(DMG_EINVAL, "dmg_ocean_jacobi is the single-band form; use dmg_ocean_jacobi_sweeps + dmg_ocean_halo");
    if (zero_first) { int rc = dmg_ocean_zero_pressure(o); if (rc) return rc; }
    return dmg_ocean_jacobi_sweeps(o, iters);
}

int dmg_ocean_halo(dmg_ocean *o, int field, int side, int rows, void **send, void **recv, size_t *bytes)
{
    DMG_CHECK_ARG(o != nullptr && send != nullptr && recv != nullptr && bytes != nullptr);
    DMG_CHECK_ARG(side == DMG_SIDE_SOUTH || side == DMG_SIDE_NORTH);
    if (field == DMG_F_VADVECT) {
        // the deeper gather source of the next advect (dmg_ocean_advect_prepare): same shape as the other fields' slabs
        if (!o->adv_ready) return dmg_fail(DMG_ESTATE, "DMG_F_VADVECT: dmg_ocean_advect_prepare must come first");
        DMG_CHECK_ARG(rows >= 1 && rows <= o->adv_rows);
        const size_t row = (size_t)o->grid.W * 2;
        const int own = o->y1 - o->y0;
        *send = *recv = nullptr; *bytes = 0;
        if (side == DMG_SIDE_SOUTH && o->adv_gs) {
            *send = o->vadv + row * o->adv_gs; *recv = o->vadv + row * (o->adv_gs - rows); *bytes = row * rows * sizeof(float);
        } else if (side == DMG_SIDE_NORTH && o->adv_gn) {
            *send = o->vadv + row * (o->adv_gs + own - rows); *recv = o->vadv + row * (o->adv_gs + own); *bytes = row * rows * sizeof(float);
        }
        return DMG_OK;
    }
    DMG_CHECK_ARG(rows >= 1 && rows <= o->ghost);
    float *base; int comps = 1;
    switch (field) {
    case DMG_F_PRESSURE: base = o->p[o->cur]; break;
    case DMG_F_DIVW: base = o->divw; break;
    case DMG_F_TERRAIN: base = o->img; break;
    case DMG_F_V: base = o->v[o->vcur]; comps = 2; break;
    default: return dmg_fail(DMG_EINVAL, "field %d has no halo", field);
    }
    size_t row = (size_t)o->grid.W * comps;
    int own = o->y1 - o->y0;
    *send = *recv = nullptr; *bytes = 0;
    if (side == DMG_SIDE_SOUTH && o->gs) {
        *send = base + row * o->gs;                      // first `rows` owned rows
        *recv = base + row * (o->gs - rows);             // ghost rows just south of them
        *bytes = row * rows * sizeof(float);
    } else if (side == DMG_SIDE_NORTH && o->gn) {
        *send = base + row * (o->gs + own - rows);       // last `rows` owned rows
        *recv = base + row * (o->gs + own);              // ghost rows just north of them
        *bytes = row * rows * sizeof(float);
    }
    return DMG_OK;
}

int dmg_ocean_get(dmg_ocean *o, float *vx, float *vy, float *pressure, float *divw)
{
    DMG_CHECK_ARG(o != nullptr);
    DMG_HIP(hipSetDevice(o->ctx->device));
    int rc;
    if (pressure && (rc = no_pending(o, "dmg_ocean_get(pressure)"))) return rc;
    if (pressure && (rc = ocean_download_rows(o, pressure, o->p[o->cur], 1))) return rc;
    if ((rc = ocean_deep_check(o, o->ctx->stream))) return rc;         // (no deep launch so far: nothing to look at)
    if (divw && (rc = ocean_download_rows(o, divw, o->divw, 1))) return rc;
    if (vx || vy) {
        size_t own = (size_t)(o->y1 - o->y0) * o->grid.W;
        std::vector<float> tmp(own * 2);
        DMG_HIP(hipMemcpyAsync(tmp.data(), o->v[o->vcur] + (size_t)o->gs * o->grid.W * 2, own * 8,
                               hipMemcpyDeviceToHost, o->ctx->stream));
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));
        size_t off = (size_t)o->y0 * o->grid.W;
        // scratch1 := v.x, scratch2 := v.y (OceanCurrents.cpp:175-205)
        if (vx) for (size_t i = 0; i < own; i++) vx[off + i] = tmp[2 * i];
        if (vy) for (size_t i = 0; i < own; i++) vy[off + i] = tmp[2 * i + 1];
    }
    return DMG_OK;
}

}  // extern "C"
