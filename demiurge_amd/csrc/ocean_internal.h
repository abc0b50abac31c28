// dmg_ocean: device state of one OceanCurrents instance (one latitude band).
#pragma once
#include <map>
#include <tuple>
#include <utility>
#include "dmg_internal.h"

// per-local-row tables, evaluated on the host in fp32 with the GLSL's own operation order
enum {
    RT_JCX = 0,     // jacobi pixelwidth2.x = (pixelsize.x/420)^2            OceanCurrents.cpp:478
    RT_JCY,         // jacobi pixelwidth2.y
    RT_JBETA,       // 2*(1/pw2.x + 1/pw2.y)                                  :479
    RT_DX,          // pixelsize(st).x  (km)                                  Shader.h:71-74
    RT_DY,          // pixelsize(st).y
    RT_DIFPX,       // diffusion 1*1/pixelsize.x^2*420*420                    :689
    RT_DIFPY,
    RT_DIFBETA,     // 2*(px+py)*(1+1/(2*(px+py)))                            :690
    RT_LAT,         // tex_to_spheric(st).y
    RT_JRX,         // RN(1/pw2.x), RN(1/pw2.y), RN(1/Beta): reciprocals for the FMA-corrected division
    RT_JRY,
    RT_JRB,
    RT_DIFRB,       // RN(1/Beta) of the diffusion sweep, for the blocked kernel's FMA-corrected division
    ROWTAB_FIELDS
};

struct dmg_ocean {
    dmg_ctx *ctx = nullptr;
    dmg_grid grid{};
    dmg_topology topo{};
    dmg_ocean_params prm{};
    int rank = 0, count = 1, ghost = 1;
    int y0 = 0, y1 = 0;            // owned grid rows [y0, y1)
    int gs = 0, gn = 0;            // ghost rows towards south / north neighbour band
    int L = 0;                     // local rows = gs + (y1-y0) + gn
    float *p[2] = { nullptr, nullptr };
    int cur = 0;                   // p[cur] holds the current pressure
    float *divw = nullptr, *img = nullptr;
    unsigned char *code = nullptr;    // Jacobi/diffusion/subDiv 'use centre' bits + centre-solid bit
    unsigned char *code8 = nullptr;   // 8-neighbour solid bits (divergence, subDiv)
    unsigned char *rowclass = nullptr; // blocked Jacobi: class of every (tile column, row) wave-row (ocean_jacobi_tb.hip)
    std::vector<unsigned char> rowclass_host;   // the same table on the host, for the live-tile lists
    struct tile_list {
        int *d = nullptr; int n = 0;
        std::vector<int> host;                           // the list as the kernel reads it (kept only for dmg_stream_units_probe)
        int twin = 0;                                    // streaming kernel: the list holds pairs of twin units (two entries per workgroup, k_jacobi_stream_twin)
        size_t n_units = 0;                              // entries in the list (2 n for twin units)
        size_t n_cut_units = 0;                          // ... that are units of the cut (not place holders)
        int cut_twin = 0;
        int n_flip = 0;                                  // streaming kernel: units of this list that stream southwards (they need the mirrored row tables)
        int cut_seg = 0, cut_seg2 = 0, cut_split = 0;   // streaming kernel: the cut of a launch's rows into units (the halves of a split launch share the whole launch's)
        long long cut_rows = 0;                          // rows its units write, summed (the halves of a split launch are checked against their cut)
    };
    // (out_lo, out_hi, rows per wave, part, zone_lo, zone_hi) of a launch -> tiles that are not land everywhere; part 0 = all
    // tile rows, 1 = tile rows touching the send zones [out_lo, zone_lo) / [zone_hi, out_hi), 2 = the tile rows in between
    std::map<std::tuple<int, int, int, int, int, int>, tile_list> tb_lists;
    bool lists_on_host = false;    // dmg_stream_units_probe: unit lists are kept in tile_list::host and never copied to a device
    // per launch geometry: which of the three candidate shapes is fastest.  A trial is timed with an event pair that is
    // read back when the geometry is launched the next time, so that trial launches run back to back with their
    // neighbours like every other launch (timing them in isolation, with a host synchronisation after each, favoured
    // shapes that lose in steady state)
    struct tb_tune { int trials[6] = { 0, 0, 0, 0, 0, 0 }; float best_ms[6] = { 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f }; int chosen = -1;
                     hipEvent_t ev0 = nullptr, ev1 = nullptr; int pending = -1; int warm = 0; };
    std::map<std::pair<int, int>, tb_tune> tb_tuning;           // (out_lo, out_hi) -> which tile shape is fastest there
    std::map<std::pair<int, int>, tb_tune> st_tuning;           // (out_lo, out_hi) -> which cut of the streaming kernel's units (ocean_jacobi_stream.hip)
    float *v[2] = { nullptr, nullptr };   // interleaved (vx,vy)
    int vcur = 0;
    // dmg_ocean_jacobi_sweeps_begin/_end: the interior part of a block's last launch, still to be issued
    struct { bool active = false; int lo = 0, hi = 0, n = 0, zone_lo = 0, zone_hi = 0; } pending;
    hipEvent_t ev_fork = nullptr;  // ocean_jacobi_sweeps_begin with an edge stream: the launch before the split one is done
    hipEvent_t ev_edge = nullptr;  // ... the edge units of the split launch are done (the interior units start behind them: edge first)
    bool edge_wait = false;
    int *oob_flag = nullptr;       // device word: advect out-of-band flag / reduction scratch
    // advect with a back-trace longer than `ghost` rows (dmg_ocean_advect_prepare): the gather source of this step, v of rows
    // [y0 - adv_gs, y1 + adv_gn) -- own rows copied from v, the rows beyond them exchanged as field DMG_F_VADVECT
    float *vadv = nullptr;
    size_t vadv_cap = 0;           // bytes allocated
    int adv_rows = 0, adv_gs = 0, adv_gn = 0;
    bool adv_ready = false;        // prepared for the next dmg_ocean_advect (which clears it)
    float *rowtab = nullptr;       // ROWTAB_FIELDS arrays of L floats
    float *jcoefp = nullptr;       // [L + 32][4] = cx, 1/cx, Beta, 1/Beta of local rows -16 .. L+15, rows beyond the grid mirrored (scalar loads, no mapping)
    float *jcoef = nullptr;        // [L][8] = cx, 1/cx, Beta, 1/Beta, cy, 1/cy, -, -: a row's Jacobi coefficients in one 16-byte load (ocean_jacobi_stream.hip)
    float *jcoef2 = nullptr;       // [L][8] = the same rows as exact two-operation divisors: (h, l) of cx, of Beta, of cy with x / c == fma(x, h, x * l) (div2.hip)
    unsigned char *rowclass_d = nullptr;       // the deep form's strips (224 columns written, 16 of halo): their class table, per terrain (ocean_jacobi_deep.hip)
    std::vector<unsigned char> rowclass_d_host;
    unsigned *deep_err = nullptr;              // the deep form's error word (a wave gave up waiting for its partner)
    float *jcoef3 = nullptr;       // [L][8] = ah, bh, c, e: the tolerance rendering's fused coefficients (ocean_tables.hip; DMG_QUIRK_JACOBI_TOLERANCE)
    // the streaming kernel's mirrored view of the band (units that stream southwards): the row tables reversed, built on first use
    float *jcoef_f = nullptr, *jcoef2_f = nullptr, *jcoef3_f = nullptr;
    unsigned char *rowbad_f = nullptr, *rowclass_f = nullptr;
    std::vector<unsigned char> rowbad_host;   // the same flags on the host (empty: no row is flagged), for the unit lists
    unsigned char *rowbad = nullptr; // [L] 1 = no exact pair for the row's cx or Beta: jcoef2 holds (c, 1/c) there and the kernel takes the three-operation form for that row
    bool div2_ok = false;          // cy has an exact pair (the rows' own divisors are handled row by row)
    int div2_nbad = 0;             // rows flagged in rowbad
    const float *jac_cx = nullptr, *jac_cy = nullptr, *jac_beta = nullptr;
    const float *jac_rx = nullptr, *jac_ry = nullptr, *jac_rb = nullptr;
    bool have_terrain = false;
    long long n_launches = 0, n_sweeps = 0;   // Jacobi statistics
    const char *last_kernel = "none";         // the kernel the most recent Jacobi launch used (bench.py names it in its roofline block)
    long long n_outer = 0;                    // outer steps completed (dmg_ocean_step / the multi-GPU driver); travels with checkpoints
    const float *rt(int field) const { return rowtab + (size_t)field * L; }
};

// between dmg_ocean_jacobi_sweeps_begin and _end the pressure buffers are half a launch apart
static inline int no_pending(const dmg_ocean *o, const char *what)
{
    if (o->pending.active)
        return dmg_fail(DMG_ESTATE, "%s between dmg_ocean_jacobi_sweeps_begin and dmg_ocean_jacobi_sweeps_end", what);
    return DMG_OK;
}

int ocean_upload_row_tables(dmg_ocean *o);
int ocean_jacobi_sweeps_begin(dmg_ocean *o, int n, hipStream_t edge_stream);   // dmg_ocean_jacobi_sweeps_begin with the edge units on a stream of their own (ocean.hip)
int ocean_build_codes(dmg_ocean *o);
int ocean_build_codes8(dmg_ocean *o);
int ocean_zero_land(dmg_ocean *o);            // p := +0 on land in BOTH ping-pong buffers (invariant of the blocked kernel)
// ocean_jacobi_tb.hip: n (1..8) sweeps over local rows [lo, hi); part/zone: see tb_lists (part 0: zones ignored)
int ocean_launch_tb(dmg_ocean *o, int lo, int hi, int n, int part = 0, int zone_lo = 0, int zone_hi = 0, int src = -1,
                    hipStream_t stream = nullptr);   // src >= 0: from p[src] into p[src ^ 1] instead of the current pair; stream: instead of the context's
int ocean_tb_tile_rows(dmg_ocean *o, int lo, int hi);    // output rows per tile row of the shape an 8-sweep launch over [lo, hi) uses; 0 while its tuner is open
// ocean_jacobi_stream.hip: 8 sweeps over local rows [lo, hi) by the streaming kernel; seg = rows per unit (0 = default)
// (variant: unused since round 5 retired the narrow and the persistent form)
int ocean_launch_stream(dmg_ocean *o, int lo, int hi, int part, int zone_lo, int zone_hi, int from, hipStream_t st, int seg = 0, int variant = -1);
bool ocean_stream_enabled();
// ocean_jacobi_deep.hip: 16 sweeps over local rows [lo, hi) by the deep form (two waves per unit, half the HBM bytes per sweep)
bool ocean_deep_enabled(const dmg_ocean *o);
int ocean_launch_deep(dmg_ocean *o, int lo, int hi, int from, hipStream_t st);
int ocean_deep_check(dmg_ocean *o, hipStream_t st);
int ocean_launch_diffusion_tb(dmg_ocean *o, int lo, int hi, int n);   // ocean_diffusion_tb.hip: n (1..8) sweeps over local rows [lo, hi)
int ocean_launch_literal(dmg_ocean *o);          // ocean_jacobi_literal.hip: one per-fragment sweep as llvmpipe evaluates the shader
bool ocean_tb_supported(const dmg_ocean *o);
bool ocean_tb_tuning_open(dmg_ocean *o, int lo, int hi);   // the shape tuner of the 8-sweep launch over [lo, hi) has not decided yet
int ocean_build_rowclass(dmg_ocean *o);
void ocean_tb_free_lists(dmg_ocean *o);
int ocean_terrain_changed(dmg_ocean *o);        // rebuilds what is derived from o->img (ocean.hip)
int ocean_upload_rows(dmg_ocean *o, float *dst, const float *src_full, int comps);
int ocean_download_rows(dmg_ocean *o, float *dst_full, const float *src, int comps);

// ---- Jacobi cell arithmetic shared by the single-sweep and the temporally blocked kernels ----
struct jacobi_row { float cx, cy, beta, rx, ry, rb; };     // pw2.x, pw2.y, Beta and their correctly rounded reciprocals

// dmg_div_rc (x / c through the row constant's reciprocal, Markstein form) lives in dmg_internal.h
__device__ __forceinline__ float jacobi_value(float pN, float pS, float pE, float pW, float b, const jacobi_row &k)
{
    float t = __fadd_rn(dmg_div_rc(__fadd_rn(pW, pE), k.cx, k.rx), dmg_div_rc(__fadd_rn(pS, pN), k.cy, k.ry));
    return dmg_div_rc(__fsub_rn(t, b), k.beta, k.rb);
}
