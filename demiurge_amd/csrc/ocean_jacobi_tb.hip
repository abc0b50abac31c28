// Temporally blocked Jacobi: up to TB_T = 8 sweeps of the pressure stencil (reference
// src/filter/OceanCurrents.cpp:450-481) per launch, bit-identical to 8 single-sweep launches.
//
// Why: one sweep per launch is HBM-bound at 13 B/cell (ocean.hip).  Jacobi updates are order-independent per
// cell, so a workgroup can load a tile plus an 8-cell halo once, advance it 8 sweeps on chip (recomputing the
// halo redundantly) and write the interior once: ~2 B per cell-step instead of 13.
//
// Tile geometry (one workgroup = NW waves):
//     extended tile 256 columns x (NW*R) rows;  wave w owns rows [R*w, R*w+R), lane j owns columns [4j, 4j+4)
//     -> each thread keeps a 4 x R block of pressure AND of the right-hand side in registers
//     output = the interior (8 cells from every edge; stale values creep inwards one cell per sweep)
// Neighbour traffic per sweep:
//     east/west : adjacent lanes of the same wave -> DPP wave_shr:1 / wave_shl:1 (no LDS)
//     north/south between the row blocks of adjacent waves -> 2 rows per wave through LDS (double-buffered by
//     sweep parity, one s_barrier per sweep)
// Per-row coefficients (6 floats: the three divisors and their reciprocals) sit in an LDS table filled once per tile.
// Canvas topology is applied when the extended tile is loaded: columns wrap (full longitude), rows beyond a pole
// are the same rows read mirrored and shifted by W/2 (offset(), src/Shader.h:86-95); rows beyond a non-pole edge
// are plain mirrors (CLAMP_TO_EDGE on a symmetric stencil).  Mirrored rows swap the N/S bits of their code.
#include "ocean_internal.h"
#include "ocean_tb.h"
#include <cstring>

struct tb_args {
    const float *pin;
    float *pout;
    const float *b;
    const unsigned char *code;
    const float *cx, *cy, *beta, *rx, *ry, *rb;    // per local row
    int W, L;
    int out_lo, out_hi;             // local rows written by this launch
    int y_first, H;                 // grid row of local row 0; grid height
    int spole, npole;
    int nsweeps;                    // 1..TB_T
    int tiles_x, tiles_y;
    const unsigned char *rowclass;  // [tiles_x][L]: 0 open ocean, 1 land everywhere, 2 mixed -- per (tile column, local row), or null
    int order;                      // tile walk: 0 natural row-major, 1 XCD-chunked 16-column groups, 2 XCD-chunked column-major
    const int *tile_list;           // tiles to run (those not land everywhere), or null = all tiles_x*tiles_y
    unsigned long long *trace;      // experiments only (DMG_TB_TRACE): per workgroup {start, loaded, swept, hw id, shader clock at loaded, at swept}; times in 100 MHz ticks
};

// Land handling for wave-rows that contain coast.  Inside the tile every set neighbour bit means "that neighbour is
// land" (the clamp-onto-self bits of non-pole edge rows are cleared at load: the mirrored row holds the cell's own
// value), and land cells hold exactly +0.0 (k_zero_land on input, this kernel on output).  Replacing a land neighbour
// by the centre value (OceanCurrents.cpp:470-473) is therefore  pN | (pC & mask)  on the bit patterns: v_bfe_i32 for
// the mask, one v_and_or_b32 for the select -- no lane masks, no VCC traffic.
// (inline asm: left to itself the compiler turns the pattern back into v_cndmask and hoists 40 lane masks per row
// out of the sweep loop into VGPR lanes, paying two v_readlane + s_nop per select)
// `tick` is the sweep counter: an unused asm input that keeps the mask extraction inside the sweep loop.
template <int BIT>
__device__ __forceinline__ float land_to_centre(unsigned code, int tick, float pC, float pK)
{
    float o;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(o) : "v"(pC), "v"(code_mask<BIT>(code, tick)), "v"(pK));
    return o;
}
template <int SHIFT>
__device__ __forceinline__ float tb_cell(float pC, float pN, float pS, float pE, float pW, float b, unsigned code,
                                         int tick, const jacobi_row &k)
{
    pN = land_to_centre<SHIFT + 0>(code, tick, pC, pN);
    pS = land_to_centre<SHIFT + 1>(code, tick, pC, pS);
    pE = land_to_centre<SHIFT + 2>(code, tick, pC, pE);
    pW = land_to_centre<SHIFT + 3>(code, tick, pC, pW);
    float r = jacobi_value(pN, pS, pE, pW, b, k);
    float o;                                                     // land centre -> +0.0 (:481):  r & ~mask
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(o) : "v"(code_mask<SHIFT + 4>(code, tick)), "v"(r));
    return o;
}

// maps an extended-tile row to its source: local row l, whether it is a mirrored copy, whether it is pole-shifted
struct tb_row { int l; bool flip, shift; };
__device__ __forceinline__ tb_row tb_map_row(const tb_args &a, int local_unclamped)
{
    tb_row m;
    int y = a.y_first + local_unclamped;
    m.flip = false; m.shift = false;
    if (y < 0) { y = -1 - y; m.flip = true; m.shift = a.spole != 0; }
    else if (y >= a.H) { y = 2 * a.H - 1 - y; m.flip = true; m.shift = a.npole != 0; }
    int l = y - a.y_first;
    m.l = min(max(l, 0), a.L - 1);      // rows past the stored band only feed cells that are not written
    return m;
}

// second launch-bound argument = waves per SIMD the register allocator must leave room for (R = 8: 128 VGPRs,
// i.e. two 8-wave workgroups per CU, so one workgroup's load phase overlaps the other's sweeps)
template <int R, int NW, bool TRACE = false>
__global__ __launch_bounds__(NW * 64, (R <= 10 ? 4 : (R <= 12 ? 3 : 2))) void k_jacobi_tb8(tb_args a)
{
    constexpr int EH = R * NW;                      // extended tile rows
    constexpr int OH = EH - 2 * TB_T;               // output rows per tile
    __shared__ float4 halo[2][NW][2][64];           // [sweep parity][wave][0 = row r=0, 1 = row r=R-1][lane]
    __shared__ float4 coef[EH][2];                  // per extended row: {cx, cy, beta, rx}, {ry, rb, -, -}

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = a.W;
    if (TRACE && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[6 * (size_t)blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        a.trace[6 * (size_t)blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hw;
    }

    // tile id -> (tile column, tile row).  Workgroups are dealt round-robin over the 8 XCDs (id % 8, observed, not
    // contractual: placement only affects speed).  order 1: every XCD gets a contiguous run of tiles, walked row by
    // row inside groups of 16 tile columns, so vertically adjacent tiles (which share 16 of their rows) run on the
    // same XCD shortly after one another while the tiles in flight still spread over 16 KB of each grid row
    // (a pure column-major walk puts all concurrent rows 64 KiB apart and hot-spots HBM channels).
    int tx, ty;
    {
        const int ntiles = a.tiles_x * a.tiles_y;
        int t = a.tile_list ? a.tile_list[blockIdx.x] : (int)blockIdx.x;
        if (a.order != 0) {
            int per = ntiles >> 3;                  // tiles per XCD (the remainder keeps its natural order)
            if (t < per * 8) t = (t & 7) * per + (t >> 3);
        }
        if (a.order == 1) {
            const int G = 16;
            int full = (a.tiles_x / G) * G;         // columns covered by full groups
            int ingroups = full * a.tiles_y;
            if (t < ingroups) {
                int g = t / (G * a.tiles_y), r = t - g * (G * a.tiles_y);
                ty = r / G; tx = g * G + (r - ty * G);
            } else {
                int rem = a.tiles_x - full, r = t - ingroups;
                ty = r / rem; tx = full + (r - ty * rem);
            }
        } else if (a.order == 2) { tx = t / a.tiles_y; ty = t - tx * a.tiles_y; }
        else { ty = t / a.tiles_x; tx = t - ty * a.tiles_x; }
    }
    const int X0 = tx * TB_OW;                      // first output column of the tile
    const int Y0 = a.out_lo + ty * OH;              // first output local row of the tile
    const int e0 = Y0 - TB_T + wave * R;            // unclamped local row of this wave's register row 0

    // this lane's 4 columns (wrapped), plain and pole-shifted
    int col = X0 - TB_T + 4 * lane;
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;

    if (threadIdx.x < EH) {                         // one thread per extended row fills the coefficient table
        tb_row m = tb_map_row(a, Y0 - TB_T + (int)threadIdx.x);
        coef[threadIdx.x][0] = make_float4(a.cx[m.l], a.cy[m.l], a.beta[m.l], a.rx[m.l]);
        coef[threadIdx.x][1] = make_float4(a.ry[m.l], a.rb[m.l], 0.0f, 0.0f);
    }

    float4 p[R], b[R];
    unsigned code[R];
    unsigned mixed_rows = 0, land_rows = 0;         // bit r: row r of this wave has coast / is entirely land
    // Row classes first.  A wave-row that is land everywhere needs neither its pressure (exactly +0 in both ping-pong
    // buffers, see k_zero_land) nor its right-hand side nor its codes, and is not stored either; an open-ocean
    // wave-row has all-zero codes.  Both are known per (tile column, row) from a table built with the terrain, so
    // only coast rows (and the few mirrored rows beyond an edge) wait for a dependent code load.
    unsigned need_code = (1u << R) - 1u;
    if (a.rowclass) {
        int rr = min(lane, R - 1);
        int lu = e0 + rr;                                              // unclamped local row of register row rr
        int yg = a.y_first + lu;
        unsigned cls = 2;
        if (yg >= 0 && yg < a.H && lu >= 0 && lu < a.L) cls = a.rowclass[(size_t)tx * a.L + lu];
        unsigned long long oc = __builtin_amdgcn_ballot_w64(lane < R && cls == 0);
        unsigned long long ld = __builtin_amdgcn_ballot_w64(lane < R && cls == 1);
        land_rows = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ld);
        need_code = ((1u << R) - 1u) & ~((unsigned)__builtin_amdgcn_readfirstlane((unsigned)oc) | land_rows);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        code[r] = 0;
        if (!(need_code & (1u << r))) continue;
        tb_row m = tb_map_row(a, e0 + r);
        size_t off = (size_t)m.l * W + (m.shift ? colp : col);
        unsigned c = *reinterpret_cast<const unsigned *>(a.code + off);
        if (m.flip) c = (c & 0xFCFCFCFCu) | ((c & 0x01010101u) << 1) | ((c & 0x02020202u) >> 1);   // swap N/S bits
        // a non-pole edge row's clamp bit only says "the tap lands on the cell itself"; its mirror row is in the tile
        int yg = a.y_first + e0 + r;
        if (yg == 0 && !a.spole) c &= ~0x02020202u;
        if (yg == a.H - 1 && !a.npole) c &= ~0x01010101u;
        code[r] = c;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (!(need_code & (1u << r))) continue;
        unsigned c = code[r];
        if (__builtin_amdgcn_ballot_w64((c & 0x1F1F1F1Fu) != 0)) mixed_rows |= (1u << r);
        if (__builtin_amdgcn_ballot_w64((c & 0x10101010u) != 0x10101010u) == 0) land_rows |= (1u << r);
    }
    land_rows = __builtin_amdgcn_readfirstlane(land_rows);
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (land_rows & (1u << r)) { p[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); b[r] = p[r]; continue; }
        tb_row m = tb_map_row(a, e0 + r);
        size_t off = (size_t)m.l * W + (m.shift ? colp : col);
        p[r] = *reinterpret_cast<const float4 *>(a.pin + off);
        b[r] = *reinterpret_cast<const float4 *>(a.b + off);
    }
    mixed_rows = __builtin_amdgcn_readfirstlane(mixed_rows);
    // wave-rows that are land everywhere stay exactly 0 (k_zero_land keeps land at 0 on input; :481 on output);
    // a tile that is land everywhere has nothing to sweep or store
    if (__syncthreads_and(land_rows == ((1u << R) - 1u))) return;
    if (TRACE && threadIdx.x == 0) {
        a.trace[6 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        a.trace[6 * (size_t)blockIdx.x + 4] = __builtin_amdgcn_s_memtime();
    }

    for (int s = 0; s < a.nsweeps; s++) {
        // publish this wave's first and last rows (values before the sweep), fetch the neighbours' rows
        const int par = s & 1;
        halo[par][wave][0][lane] = p[0];
        halo[par][wave][1][lane] = p[R - 1];
        __syncthreads();
        // the outermost waves read a slot of their own: those rows are halo, their values never reach the output
        float4 south = halo[par][wave > 0 ? wave - 1 : 0][1][lane];            // row below r = 0
        float4 north = halo[par][wave < NW - 1 ? wave + 1 : NW - 1][0][lane];  // row above r = R-1

        // rows closer than s+1 to the top/bottom edge of the extended tile are stale after this sweep whatever we
        // do (staleness creeps in one row per sweep) and nothing reads them for a valid output: skip them
        // ... and rows that are land everywhere are and stay exactly 0: one wave-uniform skip mask per sweep
        unsigned skip = land_rows;
        if (wave == 0) skip |= (2u << s) - 1u;                                  // rows r <= s (s <= 7)
        if (wave == NW - 1) { const int keep = R - 1 - s; skip |= keep <= 0 ? ~0u : ~((1u << keep) - 1u); }   // rows r >= R-1-s

        float4 prev = south;                        // pre-sweep values of row r-1
#pragma unroll
        for (int r = 0; r < R; r++) {
            float4 C = p[r];
            float4 N = (r < R - 1) ? p[r + 1] : north;
            float4 S = prev;
            prev = C;
            if (skip & (1u << r)) continue;
            float wl = dpp_from_prev_lane(C.w);     // west neighbour of C.x
            float er = dpp_from_next_lane(C.x);     // east neighbour of C.w
            float4 k0 = coef[wave * R + r][0], k1 = coef[wave * R + r][1];
            jacobi_row k = { k0.x, k0.y, k0.z, k0.w, k1.x, k1.y };
            float4 o;
            if (!(mixed_rows & (1u << r))) {        // wave-uniform: open ocean, no selects needed
                o.x = jacobi_value(N.x, S.x, C.y, wl, b[r].x, k);
                o.y = jacobi_value(N.y, S.y, C.z, C.x, b[r].y, k);
                o.z = jacobi_value(N.z, S.z, C.w, C.y, b[r].z, k);
                o.w = jacobi_value(N.w, S.w, er, C.z, b[r].w, k);
            } else {
                unsigned c = code[r];
                o.x = tb_cell<0>(C.x, N.x, S.x, C.y, wl, b[r].x, c, s, k);
                o.y = tb_cell<8>(C.y, N.y, S.y, C.z, C.x, b[r].y, c, s, k);
                o.z = tb_cell<16>(C.z, N.z, S.z, C.w, C.y, b[r].z, c, s, k);
                o.w = tb_cell<24>(C.w, N.w, S.w, er, C.z, b[r].w, c, s, k);
            }
            p[r] = o;
        }
    }

    if (TRACE && threadIdx.x == 0) {
        a.trace[6 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        a.trace[6 * (size_t)blockIdx.x + 5] = __builtin_amdgcn_s_memtime();
    }
    // store the interior: extended rows [T, EH-T), lanes 2..61, clipped to the launch's row range and to W
    const int xcol = X0 - TB_T + 4 * lane;          // unwrapped column
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
#pragma unroll
    for (int r = 0; r < R; r++) {
        int e = wave * R + r;
        int l = e0 + r;                             // the row itself (mirrored / clamped copies fail the range test)
        bool row_ok = (e >= TB_T) && (e < EH - TB_T) && (l >= a.out_lo) && (l < a.out_hi);
        if (row_ok && lane_ok && !(land_rows & (1u << r)))
        {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f q = { p[r].x, p[r].y, p[r].z, p[r].w };
            __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(a.pout + (size_t)l * W + xcol));   // never re-read in this launch
        }
    }
}

// Class of every (tile column, local row) wave-row for the tile geometry above: columns [240*tx - 8, +256) wrapped.
// Uses the codes as the kernel sees them on a real (not mirrored) row, i.e. with the clamp bit of a non-pole edge row
// cleared.  One wave per (tx, l).
__global__ __launch_bounds__(64) void k_tb_rowclass(const unsigned char *__restrict__ code, unsigned char *__restrict__ cls, int W, int L,
                                                    int y_first, int H, int spole, int npole)
{
    const int lane = threadIdx.x, tx = blockIdx.x, l = blockIdx.y;
    int col = tx * TB_OW - TB_T + 4 * lane;
    col %= W; if (col < 0) col += W;
    unsigned c = *reinterpret_cast<const unsigned *>(code + (size_t)l * W + col);
    int yg = y_first + l;
    if (yg == 0 && !spole) c &= ~0x02020202u;
    if (yg == H - 1 && !npole) c &= ~0x01010101u;
    bool any = __builtin_amdgcn_ballot_w64((c & 0x1F1F1F1Fu) != 0) != 0;
    bool all_land = __builtin_amdgcn_ballot_w64((c & 0x10101010u) != 0x10101010u) == 0;
    if (lane == 0) cls[(size_t)tx * L + l] = all_land ? 1 : (any ? 2 : 0);
}


void ocean_tb_free_lists(dmg_ocean *o)
{
    for (auto &kv : o->tb_lists)
        if (kv.second.d) (void)hipFree(kv.second.d);
    o->tb_lists.clear();
    // (the streaming kernel's mirrored row tables: the classes change with the terrain, the lists go with it)
    for (void *q : { (void *)o->jcoef_f, (void *)o->jcoef2_f, (void *)o->jcoef3_f, (void *)o->rowbad_f, (void *)o->rowclass_f }) if (q) (void)hipFree(q);
    o->jcoef_f = o->jcoef2_f = o->jcoef3_f = nullptr; o->rowbad_f = o->rowclass_f = nullptr;
    if (o->rowclass_d) { (void)hipFree(o->rowclass_d); o->rowclass_d = nullptr; o->rowclass_d_host.clear(); }
    for (auto &kv : o->tb_tuning) {
        if (kv.second.ev0) (void)hipEventDestroy(kv.second.ev0);
        if (kv.second.ev1) (void)hipEventDestroy(kv.second.ev1);
    }
    o->tb_tuning.clear();
    for (auto &kv : o->st_tuning) {
        if (kv.second.ev0) (void)hipEventDestroy(kv.second.ev0);
        if (kv.second.ev1) (void)hipEventDestroy(kv.second.ev1);
    }
    o->st_tuning.clear();
}

int ocean_build_rowclass(dmg_ocean *o)
{
    if (!ocean_tb_supported(o)) return DMG_OK;
    const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW;
    if (!o->rowclass) DMG_HIP(hipMalloc((void **)&o->rowclass, (size_t)tiles_x * o->L));
    hipLaunchKernelGGL(k_tb_rowclass, dim3(tiles_x, o->L), dim3(64), 0, o->ctx->stream, o->code, o->rowclass, o->grid.W, o->L,
                       o->y0 - o->gs, o->grid.H, o->topo.spole ? 1 : 0, o->topo.npole ? 1 : 0);
    DMG_HIP(hipGetLastError());
    ocean_tb_free_lists(o);                              // lists of the previous terrain
    o->rowclass_host.resize((size_t)tiles_x * o->L);
    DMG_HIP(hipMemcpyAsync(o->rowclass_host.data(), o->rowclass, o->rowclass_host.size(), hipMemcpyDeviceToHost, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    return DMG_OK;
}

// Tiles of a launch over local rows [lo, hi) that have any work: a tile whose extended rows are all "land everywhere"
// returns before it loads or stores anything, so it need not be dispatched at all (30 % of the tiles on the survey
// terrain; each would hold a workgroup slot for a few microseconds).  Rows beyond the stored band or the grid count as
// not-land, as in the kernel.  Built on the host from the class table, once per (lo, hi) and terrain.
template <int R, int NW>
static const dmg_ocean::tile_list *tb_live_tiles(dmg_ocean *o, int lo, int hi, int tiles_x, int tiles_y, int part, int zone_lo, int zone_hi)
{
    auto key = std::make_tuple(lo, hi, R, part, zone_lo, zone_hi);
    auto it = o->tb_lists.find(key);
    if (it != o->tb_lists.end()) return &it->second;
    constexpr int EH = R * NW, OH = EH - 2 * TB_T;
    const int y_first = o->y0 - o->gs, L = o->L, H = o->grid.H;
    const bool classes = !o->rowclass_host.empty();
    std::vector<int> live;
    live.reserve((size_t)tiles_x * tiles_y);
    for (int ty = 0; ty < tiles_y; ty++) {
        if (part) {                                       // tile rows that produce rows of a send zone vs the rest
            const int r0 = lo + ty * OH, r1 = std::min(r0 + OH, hi);
            const bool edge = r0 < zone_lo || r1 > zone_hi;
            if (edge != (part == 1)) continue;
        }
        for (int tx = 0; tx < tiles_x; tx++) {
            bool dead = classes;
            if (classes) {
                const unsigned char *cls = o->rowclass_host.data() + (size_t)tx * L;
                const int e0 = lo + ty * OH - TB_T;
                for (int e = 0; e < EH && dead; e++) {
                    int lu = e0 + e, yg = y_first + lu;
                    dead = yg >= 0 && yg < H && lu >= 0 && lu < L && cls[lu] == 1;
                }
            }
            if (!dead) live.push_back(ty * tiles_x + tx);
        }
    }
    dmg_ocean::tile_list tl;
    tl.n = (int)live.size();
    if (tl.n && hipMalloc((void **)&tl.d, live.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (tl.n && hipMemcpy(tl.d, live.data(), live.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return &(o->tb_lists[key] = tl);
}

// tile shapes built into the library: R rows per thread x NW waves per workgroup.  The default is chosen from
// measurements; DMG_TB_SHAPE=<R>x<NW> overrides it for experiments.
struct tb_shape { int R, NW; };
#define TB_NCAND 5
static tb_shape tb_forced_shape()
{
    static const tb_shape forced = [] {
        tb_shape d = { 0, 0 };
        const char *e = dmg_hook("DMG_TB_SHAPE");
        int r = 0, nw = 0;
        // stream: the streaming kernel for every 8-sweep launch
        if (e && !strcmp(e, "stream")) d = { -1000, 0 };
        else if (e && sscanf(e, "%dx%d", &r, &nw) == 2 &&
            ((r == 8 && nw == 8) || (r == 16 && nw == 8) || (r == 8 && nw == 16) || (r == 12 && nw == 8) || (r == 10 && nw == 8) ||
             (r == 4 && nw == 8) || (r == 6 && nw == 8)))
            d = { r, nw };
        return d;
    }();
    return forced;
}

template <int R, int NW>
static int tb_launch(dmg_ocean *o, tb_args a, int W, int rows, hipStream_t stream, dmg_ocean::tb_tune *timed = nullptr, int part = 0,
                     int zone_lo = 0, int zone_hi = 0)
{
    constexpr int OH = R * NW - 2 * TB_T;
    a.tiles_x = (W + TB_OW - 1) / TB_OW;
    a.tiles_y = (rows + OH - 1) / OH;
    static const int order = [] { const char *e = dmg_hook("DMG_TB_ORDER"); return e ? atoi(e) : 0; }();
    a.order = order;
    a.trace = nullptr;
    a.tile_list = nullptr;
    size_t nt = (size_t)a.tiles_x * a.tiles_y;
    static const bool all_tiles = dmg_hook("DMG_TB_ALLTILES") != nullptr;
    if ((a.rowclass && order == 0 && !all_tiles) || part) {
        const dmg_ocean::tile_list *tl = tb_live_tiles<R, NW>(o, a.out_lo, a.out_hi, a.tiles_x, a.tiles_y, part, zone_lo, zone_hi);
        if (!tl && part) return dmg_fail(DMG_ENOMEM, "tile list allocation failed");
        if (tl) {                                         // (no list for a whole launch: every tile runs)
            if (tl->n == 0) return DMG_OK;                // land everywhere, or an empty part: nothing to do
            a.tile_list = tl->d;
            nt = (size_t)tl->n;
        }
        if (part) a.order = 0;                            // lists of a part hold natural tile ids
    }
    static const char *trace_path = dmg_hook("DMG_TB_TRACE");      // experiments: dump the per-workgroup timeline of every launch
    if (trace_path) {
        (void)hipMalloc((void **)&a.trace, nt * 48);
        (void)hipMemsetAsync(a.trace, 0, nt * 48, stream);
    }
    if (timed) (void)hipEventRecord(timed->ev0, stream);
    if (trace_path && R == 8 && NW == 8)       // the timeline build exists for the default shape only
        hipLaunchKernelGGL((k_jacobi_tb8<8, 8, true>), dim3((unsigned)nt), dim3(NW * 64), 0, stream, a);
    else
        hipLaunchKernelGGL((k_jacobi_tb8<R, NW>), dim3((unsigned)nt), dim3(NW * 64), 0, stream, a);
    if (timed) (void)hipEventRecord(timed->ev1, stream);      // read back at the next launch of this geometry
    if (trace_path) {
        std::vector<unsigned long long> h(nt * 6);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), a.trace, nt * 48, hipMemcpyDeviceToHost);
        (void)hipFree(a.trace);
        if (FILE *f = fopen(trace_path, "wb")) { fwrite(h.data(), 48, nt, f); fclose(f); }
    }
    return DMG_OK;
}

// host launcher: `n` (1..8) sweeps over local rows [lo, hi).  The caller flips o->cur.
// part 1 / 2 launch only the tile rows that do / do not produce rows of [lo, zone_lo) or [zone_hi, hi): the two parts of
// one tiling, so that the rows a neighbour band waits for can be sent while the rest is still being swept.
int ocean_launch_tb(dmg_ocean *o, int lo, int hi, int n, int part, int zone_lo, int zone_hi, int src, hipStream_t stream)
{
    if (hi <= lo) return DMG_OK;
    tb_args a;
    const int from = src >= 0 ? (src & 1) : o->cur;
    a.pin = o->p[from];
    a.pout = o->p[from ^ 1];
    a.b = o->divw;
    a.code = o->code;
    a.cx = o->jac_cx; a.cy = o->jac_cy; a.beta = o->jac_beta;
    a.rx = o->jac_rx; a.ry = o->jac_ry; a.rb = o->jac_rb;
    a.W = o->grid.W; a.L = o->L;
    a.out_lo = lo; a.out_hi = hi;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    a.nsweeps = n;
    static const bool use_cls = dmg_hook("DMG_TB_NOCLASS") == nullptr;
    a.rowclass = use_cls ? o->rowclass : nullptr;
    hipStream_t st = stream ? stream : o->ctx->stream;
    // Tile shape.  Every shape gives bit-identical results, so which one runs is purely a matter of time.  Large
    // launches (thousands of tiles) keep the chip full with 64-row tiles.  Smaller ones are bound by the latency of a
    // tile or by how the tile count divides into the 512-768 workgroup slots; there the first few launches of a
    // geometry time the three candidates (two launches each) and the rest use the fastest:
    // 1024x512 picks 4x8 (17.2 -> 10.3 ms per 5000-sweep solve), 4096x2048 picks 6x8 (46.0 -> 40.7 ms).
    tb_shape sh = tb_forced_shape();
    dmg_ocean::tb_tune *tune = nullptr, *timed = nullptr;
    int cand = -1;
    // candidates: three tile shapes of the workgroup kernel (k_jacobi_tb8) and the streaming kernel (k_jacobi_stream,
    // ocean_jacobi_stream.hip) with its own choice of unit height and with short units.  R < 0 = streaming with -R rows per unit.
    static const tb_shape cands[TB_NCAND] = { { 8, 8 }, { 6, 8 }, { 4, 8 }, { -1000, 0 }, { -32, 0 } };   // -1000: unit height from the dispatch simulation
    const bool stream_ok = n == TB_T && ocean_stream_enabled();
    if (sh.R < 0 && !stream_ok) sh = tb_shape{ 8, 8 };      // (a forced streaming form: launches of fewer than 8 sweeps keep the tile kernel)
    if (!sh.R) {
        const int tiles88 = ((a.W + TB_OW - 1) / TB_OW) * ((hi - lo + 47) / 48);
        // thousands of tiles: the chip is full either way and the streaming kernel does the least work
        sh = stream_ok ? tb_shape{ -1000, -1 } : tb_shape{ 8, 8 };   // -1000: the streaming kernel's default unit height
        // (a band of a split grid does not time candidates: its launches share the GPU with its second stream's edge units
        // and exchanges -- and on a one-GPU rehearsal with the other bands -- so the timings mean nothing; and the tile shapes
        // only win on grids too small to be worth splitting)
        // (the tolerance rendering is the streaming kernel's: every 8-sweep launch takes it, whatever the grid's size)
        if (tiles88 <= 4096 && n == TB_T && a.rowclass && o->count == 1 && !(stream_ok && (o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE))) {
            tune = &o->tb_tuning[std::make_pair(lo, hi)];
            if (tune->chosen < 0 && !part) {
                if (!tune->ev0 && (hipEventCreate(&tune->ev0) != hipSuccess || hipEventCreate(&tune->ev1) != hipSuccess))
                    return dmg_fail(DMG_EHIP, "hipEventCreate failed");
                if (tune->pending >= 0) {                      // the previous trial of this geometry: finished long ago
                    float ms = 1e30f;
                    (void)hipEventSynchronize(tune->ev1);
                    if (hipEventElapsedTime(&ms, tune->ev0, tune->ev1) != hipSuccess) ms = 1e30f;
                    tune->trials[tune->pending]++;
                    if (ms < tune->best_ms[tune->pending]) tune->best_ms[tune->pending] = ms;
                    tune->pending = -1;
                }
                const int ncand = stream_ok ? TB_NCAND : 3;
                cand = 0;
                while (cand < ncand && tune->trials[cand] >= 2) cand++;
                if (cand == ncand) {
                    tune->chosen = 0;
                    for (int c = 1; c < ncand; c++) if (tune->best_ms[c] < tune->best_ms[tune->chosen]) tune->chosen = c;
                    cand = -1;
                } else { timed = tune; tune->pending = cand; }
            }
            // the two parts of a split launch share one tiling: the geometry's chosen shape, or the default until the
            // whole launches of this geometry have been timed
            sh = cands[cand >= 0 ? cand : (tune->chosen >= 0 ? tune->chosen : 0)];
        }
    }
    int rc;
    if (sh.R >= 0) o->last_kernel = "k_jacobi_tb8";
    if (sh.R < 0) {
        if (timed) (void)hipEventRecord(timed->ev0, st);
        rc = ocean_launch_stream(o, lo, hi, part, zone_lo, zone_hi, from, st, sh.R == -1000 ? 0 : -sh.R, sh.NW);
        if (timed) (void)hipEventRecord(timed->ev1, st);
    }
    else if (sh.R == 4) rc = tb_launch<4, 8>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    else if (sh.R == 6) rc = tb_launch<6, 8>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    else if (sh.R == 16) rc = tb_launch<16, 8>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    else if (sh.R == 10) rc = tb_launch<10, 8>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    else if (sh.R == 12) rc = tb_launch<12, 8>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    else if (sh.NW == 16) rc = tb_launch<8, 16>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    else rc = tb_launch<8, 8>(o, a, a.W, hi - lo, st, timed, part, zone_lo, zone_hi);
    if (rc) return rc;
    DMG_HIP(hipGetLastError());
    return DMG_OK;
}

// output rows per tile row (R*8 - 16) of the shape the 8-sweep launch over [lo, hi) runs with; 0 = tuner still open
int ocean_tb_tile_rows(dmg_ocean *o, int lo, int hi)
{
    tb_shape f = tb_forced_shape();
    // the streaming forms cut a launch into units of two heights, trimmed to their water rows: there is no tile-row grid
    // for a host to plan edge chains on (bands.py falls back to the plain order); _begin/_end split along unit boundaries
    if (f.R < 0) return 0;
    if (f.R) return f.R * f.NW - 2 * TB_T;
    if (ocean_tb_tuning_open(o, lo, hi)) return 0;
    static const int rows_of[TB_NCAND] = { 8 * 8 - 2 * TB_T, 6 * 8 - 2 * TB_T, 4 * 8 - 2 * TB_T, 0, 0 };   // streaming forms: no tile-row grid
    auto it = o->tb_tuning.find(std::make_pair(lo, hi));
    if (it != o->tb_tuning.end() && it->second.chosen >= 0) return rows_of[it->second.chosen];
    const int tiles88 = ((o->grid.W + TB_OW - 1) / TB_OW) * ((hi - lo + 47) / 48);
    return (tiles88 > 4096 && ocean_stream_enabled()) ? 0 : rows_of[0];
}

// true while the per-geometry shape tuner (ocean_launch_tb) would still time a candidate for a whole 8-sweep launch
bool ocean_tb_tuning_open(dmg_ocean *o, int lo, int hi)
{
    if (tb_forced_shape().R) return false;
    static const bool use_cls = dmg_hook("DMG_TB_NOCLASS") == nullptr;
    if (!use_cls || !o->rowclass || o->count > 1) return false;
    if ((o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE) && ocean_stream_enabled()) return false;
    const int tiles88 = ((o->grid.W + TB_OW - 1) / TB_OW) * ((hi - lo + 47) / 48);
    if (tiles88 > 4096) return false;
    auto it = o->tb_tuning.find(std::make_pair(lo, hi));
    return it == o->tb_tuning.end() || it->second.chosen < 0;
}

// the temporally blocked kernel needs: full longitude (columns wrap), float4-aligned pole shift, a tile-wide row
bool ocean_tb_supported(const dmg_ocean *o)
{
    return o->topo.lonwrap && (o->grid.W % 8 == 0) && o->grid.W >= TB_EW;
}
