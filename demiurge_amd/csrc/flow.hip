// Flow accumulation: FlowFilter (reference src/filter/FlowFilter.cpp) rebuilt for the GPU.
//
// The reference alternates GLSL passes with a CPU graph walk over downloaded textures (recursive std::function
// lambdas, per-lake std::set/unordered_map).  Here everything that is per-cell lives on the device:
//   k_flow_direction   stochastic-D8 direction codes            (:109-244)
//   k_flow_mask        receiver bitmask + sink / river-mouth bits (:268-303)
//   k_flow_parent + pointer jumping   catchment label = sink cell, depth = distance to the sink
//                                     (replaces the recursive flood of assignLakeIds, :360-398)
//   k_flow_passes      lowest neighbour in another catchment per boundary cell (the per-cell part of
//                      findAllConnections, :405-492), compacted to a candidate list
//   k_flow_level       upstream-area sum, one launch per depth level from the leaves down, each cell adding its
//                      upstream neighbours in the reference's order (calculateflow's recursion, :607-622) so the fp32
//                      sums are the reference's sums
// Only the lake graph (which catchment overflows into which: solvingConnections, :533-595) runs on the host, on the
// compacted candidate list.  Tie-breaking where the reference is container-order dependent follows
// oracle/flow_oracle.cpp: total order (height, from, tolocation), lakes in ascending cell index.
#include "canvas.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <queue>
#include <map>
#include <set>
#include <tuple>
#include <unordered_map>
#include <atomic>
#include <chrono>
#include <thread>
#include <sys/mman.h>
#include <cstdlib>
#include <new>

// keypad order of FlowFilter::neighbours (:66-73): bits 1,2,3,4,6,7,8,9 <-> (dx,dy)
__constant__ int c_ndx[8] = { -1, 0, 1, -1, 1, -1, 0, 1 };
__constant__ int c_ndy[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
__constant__ int c_nbit[8] = { 1, 2, 4, 8, 32, 64, 128, 256 };      // value of keypad bit 1,2,3,4,6,7,8,9

struct flow_canvas {
    canvas c;            // the project's canvas (mask pass)
    canvas cm;           // cornerCoords moved 1e-3 off the poles (direction pass, :253-256)
    int wrap;            // FlowFilter::neighbours' own wrap test (:47)
};

__device__ __forceinline__ float d_fract(float x) { return __fsub_rn(x, floorf(x)); }
// hash() and noise() of the direction shader (:114-131).  noise() is sampled at st*resolution*2 = (2x+1, 2y+1), an
// integer lattice point in exact arithmetic; in fp32 it can land an ulp below (non-power-of-two sizes), so the full
// bilinear form is kept.
// hash() as Mesa's algebraic pass leaves it (probe shader: 131072/131072 lattice points; the literal form 99.4 %): with
// p = 50*fract(..), the sum p.x + p.y is factored into 50*(fract_x + fract_y), the product p.x*p.y is not touched
__device__ __forceinline__ float d_flow_hash(float px, float py)
{
    float fx = d_fract(__fadd_rn(__fmul_rn(px, 0.3183099f), 0.71f)), fy = d_fract(__fadd_rn(__fmul_rn(py, 0.3183099f), 0.113f));
    px = __fmul_rn(50.0f, fx);
    py = __fmul_rn(50.0f, fy);
    return __fadd_rn(-1.0f, __fmul_rn(2.0f, d_fract(__fmul_rn(__fmul_rn(px, py), __fmul_rn(50.0f, __fadd_rn(fx, fy))))));
}
__device__ __forceinline__ float d_mix(float a, float b, float t)      // GLSL mix as lowered by Mesa (nir_lower_flrp): a + t*(b - a)
{
    return __fadd_rn(a, __fmul_rn(t, __fsub_rn(b, a)));
}
__device__ __forceinline__ float d_flow_noise(float px, float py)
{
    float ix = floorf(px), iy = floorf(py), fx = d_fract(px), fy = d_fract(py);
    float ux = __fmul_rn(__fmul_rn(fx, fx), __fsub_rn(3.0f, __fmul_rn(2.0f, fx)));
    float uy = __fmul_rn(__fmul_rn(fy, fy), __fsub_rn(3.0f, __fmul_rn(2.0f, fy)));
    return d_mix(d_mix(d_flow_hash(ix, iy), d_flow_hash(__fadd_rn(ix, 1.0f), iy), ux),
                 d_mix(d_flow_hash(ix, __fadd_rn(iy, 1.0f)), d_flow_hash(__fadd_rn(ix, 1.0f), __fadd_rn(iy, 1.0f)), ux), uy);
}

// ---- direction codes ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_direction(flow_canvas fc, const float *__restrict__ img, const float *__restrict__ sel,
                                                        const float *__restrict__ dxrow, const float *__restrict__ dyrow,
                                                        unsigned char *__restrict__ code)
{
    const canvas &c = fc.cm;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= c.W) return;
    size_t i = (size_t)y * c.W + x;
    float a = img[i];
    if (a <= 0.0f || (sel && sel[i] == 0.0f)) { code[i] = 0; return; }
    // get_aspect -> get_texture_gradient (reference src/Shader.h:281-297): taps a..i = -vec2(ox,oy)
    float A = img[d_step(c, x, y, -1, -1)], B = img[d_step(c, x, y, 0, -1)], C = img[d_step(c, x, y, 1, -1)];
    float D = img[d_step(c, x, y, -1, 0)], F = img[d_step(c, x, y, 1, 0)];
    float G = img[d_step(c, x, y, -1, 1)], Hh = img[d_step(c, x, y, 0, 1)], I = img[d_step(c, x, y, 1, 1)];
    float gx = __fdiv_rn(__fadd_rn(-__fadd_rn(__fadd_rn(C, __fmul_rn(2.0f, F)), I), __fadd_rn(__fadd_rn(A, __fmul_rn(2.0f, D)), G)),
                         __fmul_rn(8.0f, dxrow[y]));
    float gy = __fdiv_rn(__fsub_rn(__fadd_rn(__fadd_rn(G, __fmul_rn(2.0f, Hh)), I), __fadd_rn(__fadd_rn(A, __fmul_rn(2.0f, B)), C)),
                         __fmul_rn(8.0f, dyrow[y]));
    float aspect = __fsub_rn(DMG_PI, glsl::atan2(gy, -gx));
    const float two_pi = 2.0f * DMG_PI;
    float lower = __fmul_rn(__fmul_rn(__fdiv_rn(floorf(__fmul_rn(__fdiv_rn(aspect, two_pi), 8.0f)), 8.0f), 2.0f), DMG_PI);
    float upper = __fmul_rn(__fmul_rn(__fdiv_rn(ceilf(__fmul_rn(__fdiv_rn(aspect, two_pi), 8.0f)), 8.0f), 2.0f), DMG_PI);
    float prob = __fmul_rn(__fdiv_rn(fabsf(__fsub_rn(aspect, lower)), DMG_PI), 4.0f);
    float sx = d_stx(c, x), sy = d_sty(c, y);
    float q = __fadd_rn(__fmul_rn(d_flow_noise(__fmul_rn(__fmul_rn(sx, (float)c.W), 2.0f), __fmul_rn(__fmul_rn(sy, (float)c.H), 2.0f)), 0.5f), 0.5f);
    aspect = (q < prob) ? upper : lower;
    int dx = (int)roundf(glsl::cos(aspect)), dy = -(int)roundf(glsl::sin(aspect));
    int fcode = 5 + dx + 3 * dy;                       // keypad code of (dx,dy): 9=(1,1) 8=(0,1) .. 1=(-1,-1); (0,0) stays 5
    int n = d_step(c, x, y, dx, dy);
    float a2 = img[n];
    float s2 = sel ? sel[n] : 1.0f;
    if (a2 <= 0.0f) fcode = 5;
    if (s2 == 0.0f) fcode = 5;
    if (a2 < a) { code[i] = (unsigned char)fcode; return; }
    // fallback: the lowest of the 8 neighbours, visited in the shader's order (:134-189)
    fcode = 5;
    float s = 1.0f;
    const int ox[8] = { 1, 0, -1, 1, -1, 1, 0, -1 }, oy[8] = { 1, 1, 1, 0, 0, -1, -1, -1 }, oc[8] = { 9, 8, 7, 6, 4, 3, 2, 1 };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int m = d_step(c, x, y, ox[k], oy[k]);
        a2 = img[m];
        if (a2 < a) { fcode = oc[k]; a = a2; s = sel ? sel[m] : 1.0f; }
    }
    if (a <= 0.0f) fcode = 5;                          // edge of ocean
    if (s == 0.0f) fcode = 5;
    code[i] = (unsigned char)fcode;
}

// ---- receiver bitmask -------------------------------------------------------------------------------------------
// mask bits: 1,2,4,8,32,64,128,256 = keypad neighbours 1,2,3,4,6,7,8,9 drain into this cell; 16 = sink; 512 = river
// mouth (some neighbour is ocean/unselected); 0xFFFF = not land (the reference's -1)
__global__ __launch_bounds__(256) void k_flow_mask(flow_canvas fc, const float *__restrict__ img, const float *__restrict__ sel,
                                                   const unsigned char *__restrict__ code, unsigned short *__restrict__ mask)
{
    const canvas &c = fc.c;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= c.W) return;
    size_t i = (size_t)y * c.W + x;
    if (img[i] <= 0.0f || (sel && sel[i] == 0.0f)) { mask[i] = 0xFFFFu; return; }
    unsigned v = 0;
    bool flag = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // keypad neighbour k drains into this cell when its code points back: code == 10 - keypad(k)
        int kp = (k < 4) ? k + 1 : k + 2;
        unsigned cc = code[d_step(c, x, y, c_ndx[k], c_ndy[k])];
        if (cc == (unsigned)(10 - kp)) v |= (unsigned)c_nbit[k];
        if (cc == 0) flag = true;
    }
    if (code[i] == 5) v |= 16u;
    if (flag) v |= 512u;
    mask[i] = (unsigned short)v;
}

// ---- catchments ---------------------------------------------------------------------------------------------------
// parent = the cell this one drains to under FlowFilter::neighbours' rules (x wraps on a full-longitude canvas, a
// step out of the grid in y leads nowhere); sinks and dead ends are their own parent.
__global__ void k_flow_parent(flow_canvas fc, const unsigned char *__restrict__ code, int *__restrict__ parent, int *__restrict__ dist)
{
    int W = fc.c.W, H = fc.c.H;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H) return;
    int cd = code[i];
    int p = (int)i, d = 0;
    if (cd != 0 && cd != 5) {
        int dx = (cd - 1) % 3 - 1, dy = (cd - 1) / 3 - 1;
        int x = (int)(i % W) + dx, y = (int)(i / W) + dy;
        bool ok = true;
        if (fc.wrap) x = (x + W) % W; else if (x < 0 || x >= W) ok = false;
        if (y < 0 || y >= H) ok = false;
        if (ok) { p = y * W + x; d = 1; }
    }
    parent[i] = p;
    dist[i] = d;
}

// One round of pointer doubling, from one pair of tables into the other.  A cell whose parent is a root is done: it adds the root's 0
// and keeps its pointer -- and from the second round after that on, the table it would write already holds the same pair (each
// table is consistent by itself: a pointer and the distance to it; a pointer to a root comes with the final distance), so
// nothing is written.  After a few rounds that is nine cells in ten: 12 bytes a cell instead of 24 (round 5).  first: the
// table written has nothing in it yet.
__global__ void k_flow_jump(const int *__restrict__ pin, const int *__restrict__ din, int *pout, int *dout, size_t n, int *__restrict__ changed, int first)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = pin[i];
    if (p == (int)i && !first) return;                 // (a root -- every sea cell is one: both tables hold (i, 0) since the first round; not even read)
    int pp = pin[p];
    if (pp == p && !first && pout[i] == p) return;
    dout[i] = din[i] + din[p];
    pout[i] = pp;
    if (pp != p) *changed = 1;
}

// label = root when the root is a sink, else -1 (ocean, unselected, or a chain that leaves the grid)
__global__ void k_flow_label(const int *__restrict__ root, const unsigned char *__restrict__ code, int *__restrict__ label, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = root[i];
    label[i] = (code[i] != 0 && code[r] == 5) ? r : -1;
}

// ---- dense lake numbering -------------------------------------------------------------------------------------------
// A lake is a sink cell that labels itself.  dense[c] (exclusive scan of the flags) numbers the lakes 0..L-1 in cell
// order, so the host side of the lake graph works on flat arrays instead of hash maps keyed by cell index.
__global__ void k_flow_sinkflag(const int *__restrict__ label, int *__restrict__ flag, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (label[i] == (int)i) ? 1 : 0;
}
__global__ void k_flow_sinkcells(const int *__restrict__ label, const int *__restrict__ dense, int *__restrict__ sink_cell, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && label[i] == (int)i) sink_cell[dense[i]] = (int)i;
}

// Appending to one list from every cell: atomics on ONE address are served one wave-instruction at a time, ~5 ns each,
// whether 64 lanes or one take part (k_flow_passes with an atomicAdd per candidate: 0.47 ms at 4096x2048, 10 ms at
// 16384x8192 -- and exactly the same with one add per wave).  So a block of 256 threads looks at FLOW_ITEMS cells per
// thread, keeps what it finds in registers, and takes all its slots with a single add.  Called by every thread of the
// block, once.
#define FLOW_ITEMS 16
__device__ __forceinline__ int block_append_base(int c, int *count)
{
    __shared__ int s_tot[4], s_base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_tot[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { const int t = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3]; s_base = t ? atomicAdd(count, t) : 0; }
    __syncthreads();
    int off = s_base;
    for (int k = 0; k < w; k++) off += s_tot[k];
    return off + inc - c;
}

// ---- pass candidates --------------------------------------------------------------------------------------------
// One candidate per boundary cell s: the lowest neighbour that belongs to another catchment (:470-492).
//   key = (dense id of the lake of s) << 32 | (dense id of the neighbouring lake)     val = (pass height bits) << 32 | s
// heights are > 0, so their fp32 bit patterns order like the values; dense ids order like the sink cell indices.
// Round 6: a block takes a TILE of 256 columns x FLOW_ITEMS rows, a thread one column of it, top to bottom, with the labels of its three
// columns in the three rows around the cell in registers -- three label loads per cell, the left and right ones out of the L1, instead
// of nine from wherever the neighbouring rows' blocks had left them; and no integer division: the first version turned every cell's
// index back into (x, y) with a 64-bit `%` and `/` and wrapped each of its eight neighbours with a 32-bit `%` -- ~400 instructions of
// division per cell for a kernel that reads 12 bytes of it (5.7 ms at 16384 x 8192, profiles/r05_world_kernels.txt).
__global__ __launch_bounds__(256) void k_flow_passes(flow_canvas fc, const float *__restrict__ h, const int *__restrict__ label, const int *__restrict__ dense,
                                                     const unsigned short *__restrict__ mask, unsigned long long *__restrict__ key,
                                                     unsigned long long *__restrict__ val, int *__restrict__ count, int cap)
{
    const int W = fc.c.W, H = fc.c.H;
    int found[FLOW_ITEMS];                              // the neighbour cell the pass leads to, or -1
    float pass[FLOW_ITEMS];
    int c = 0;
    const int x0 = (int)blockIdx.x * 256 + (int)threadIdx.x, yb = (int)blockIdx.y * FLOW_ITEMS;
    const bool col_ok = x0 < W;
    // the three columns: the cell's own, its western and its eastern neighbour's (wrapped, or absent: -1)
    int xw = x0 - 1, xe = x0 + 1;
    if (fc.wrap) { if (xw < 0) xw += W; if (xe >= W) xe -= W; }
    else { if (xe >= W) xe = -1; }
    auto row3 = [&](int y, int &lw, int &lc, int &le) {   // labels of row y at the three columns; -1 outside the canvas
        lw = lc = le = -1;
        if (col_ok && y >= 0 && y < H) {
            const int *r = label + (size_t)y * W;
            lc = r[x0];
            if (xw >= 0) lw = r[xw];
            if (xe >= 0) le = r[xe];
        }
    };
    int aw, ac, ae, bw, bc, be, nw, nc, ne;              // rows y - 1, y, y + 1
    row3(yb - 1, aw, ac, ae);
    row3(yb, bw, bc, be);
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const int y0 = yb + t;
        row3(y0 + 1, nw, nc, ne);
        const int lake = bc;                            // (-1: not a cell of the canvas, or no lake)
        float minpass = INFINITY;
        int nlake = -1;
        if (lake >= 0) {
            // FlowFilter::neighbours order (keypad 1,2,3,4,6,7,8,9 = (dx,dy): row y-1 west..east, row y west, east, row y+1 west..east);
            // strict '<' keeps the first
            const int ln[8] = { aw, ac, ae, bw, be, nw, nc, ne };
            const int nx[8] = { xw, x0, xe, xw, xe, xw, x0, xe };
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (ln[k] < 0 || ln[k] == lake) continue;   // the adjacency pass: another lake's id (:417-439)
                const int n = (y0 + c_ndy[k]) * W + nx[k];
                const float bd = h[n];
                if (bd > 0.0f && bd < minpass) { minpass = bd; nlake = n; }     // :476-481
            }
        }
        if (nlake >= 0 && (mask[label[nlake]] & 512u)) nlake = -1;             // never towards a river-mouth lake (:484)
        found[t] = nlake; pass[t] = minpass;
        c += nlake >= 0;
        aw = bw; ac = bc; ae = be; bw = nw; bc = nc; be = ne;
    }
    int slot = block_append_base(c, count);             // (the order of the candidates does not matter: they are sorted next)
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        if (found[t] < 0) continue;
        const size_t i = (size_t)(yb + t) * W + x0;
        if (slot < cap) {
            key[slot] = ((unsigned long long)(unsigned)dense[label[i]] << 32) | (unsigned)dense[label[found[t]]];
            val[slot] = ((unsigned long long)__float_as_uint(fmaxf(pass[t], h[i])) << 32) | (unsigned)i;
        }
        slot++;
    }
}

// per unique (lake, from) pass: sort key (lake << 32 | height bits), payload (from << 32 | s), depth of s
__global__ void k_flow_pack(const unsigned long long *ukey, const unsigned long long *uval, unsigned long long *key2, unsigned long long *pay, int n)      // (in place)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = ukey[i], v = uval[i];
    key2[i] = (k & 0xFFFFFFFF00000000ull) | (v >> 32);
    pay[i] = (k << 32) | (v & 0xFFFFFFFFull);
}

__global__ void k_flow_gather_dist(const unsigned long long *__restrict__ pay, const int *__restrict__ dist, int *__restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dist[(unsigned)(pay[i] & 0xFFFFFFFFull)];
}

// river-mouth lakes = sinks (bit 16) next to the ocean (bit 512), compacted
__global__ __launch_bounds__(256) void k_flow_mouths(const unsigned short *__restrict__ mask, const int *__restrict__ dense, int *__restrict__ out,
                                                     int *__restrict__ count, size_t n)
{
    unsigned hit = 0;
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const size_t i = ((size_t)blockIdx.x * FLOW_ITEMS + t) * 256 + threadIdx.x;
        const unsigned m = i < n ? mask[i] : 0u;
        if (m != 0xFFFFu && (m & 16u) && (m & 512u)) hit |= 1u << t;
    }
    int slot = block_append_base(__popc(hit), count);
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++)
        if (hit & (1u << t)) out[slot++] = dense[((size_t)blockIdx.x * FLOW_ITEMS + t) * 256 + threadIdx.x];
}

// is there a cell that could become sea -- not safely above zero -- at all?
__global__ __launch_bounds__(256) void k_flow_any_sea(const float *__restrict__ h, int *__restrict__ found, size_t n)
{
    bool hit = false;
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const size_t i = ((size_t)blockIdx.x * FLOW_ITEMS + t) * 256 + threadIdx.x;
        if (i < n) hit = hit || !(h[i] >= 1e-20f);          // (NaN counts)
    }
    if (__builtin_amdgcn_ballot_w64(hit) && (threadIdx.x & 63) == 0) *found = 1;
}
// is there a river mouth at all?  (k_flow_mouths' test, counted only)
__global__ __launch_bounds__(256) void k_flow_any_mouth(const unsigned short *__restrict__ mask, int *__restrict__ found, size_t n)
{
    bool hit = false;
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const size_t i = ((size_t)blockIdx.x * FLOW_ITEMS + t) * 256 + threadIdx.x;
        const unsigned m = i < n ? mask[i] : 0u;
        hit = hit || (m != 0xFFFFu && (m & 16u) && (m & 512u));
    }
    if (__builtin_amdgcn_ballot_w64(hit) && (threadIdx.x & 63) == 0) *found = 1;
}

__global__ void k_flow_scatter_pairs(const int *__restrict__ idx, const int *__restrict__ v, int *__restrict__ dst, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = v[i];
}

// the reduced pass list as the host's growth wants it: 16 bytes per entry {height, from, s, lake} and per lake the range of
// its entries (sorted by (lake, height, from)); a lake without passes keeps {0, 0}
__global__ void k_flow_entries(const unsigned long long *__restrict__ key, const unsigned long long *__restrict__ pay, int4 *__restrict__ ent,
                               int2 *__restrict__ span, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lake = (int)(key[i] >> 32);
    ent[i] = make_int4((int)(unsigned)key[i], (int)(pay[i] >> 32), (int)(unsigned)pay[i], lake);
    if (i == 0 || (int)(key[i - 1] >> 32) != lake) span[lake].x = i;
    if (i == n - 1 || (int)(key[i + 1] >> 32) != lake) span[lake].y = i + 1;
}
// connections[tolocation] = pass (FlowFilter.cpp:588): of the passes assigned to one cell the LAST one survives.
// assigned[j] = entry index of the j-th connection the growth made
__global__ void k_flow_lastpos(const int *__restrict__ assigned, const unsigned long long *__restrict__ pay, int *__restrict__ lastpos, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) atomicMax(&lastpos[(unsigned)pay[assigned[j]]], j);
}
// the surviving connections: the cell drains into the connected lake's sink cell, the lake hangs below the lake of that cell
__global__ void k_flow_connect(const int *__restrict__ assigned, const unsigned long long *__restrict__ key, const unsigned long long *__restrict__ pay,
                               const int *__restrict__ sdist, const int *__restrict__ lastpos, const int *__restrict__ sink_cell,
                               int *__restrict__ conn_from, int *__restrict__ lpar, int *__restrict__ loff, float *__restrict__ llvl, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int c = assigned[j];
    const unsigned s = (unsigned)pay[c];
    if (lastpos[s] != j) return;
    const int from = (int)(pay[c] >> 32);
    conn_from[s] = sink_cell[from];
    lpar[from] = (int)(key[c] >> 32);
    loff[from] = sdist[c] + 1;
    llvl[from] = __uint_as_float((unsigned)key[c]);
}
__global__ void k_flow_lake_init(int *__restrict__ lpar, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lpar[i] = i;
}
__global__ void k_flow_mark(const int *__restrict__ idx, unsigned char *__restrict__ flag, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[idx[i]] = 1;
}

// ---- lake tree: depth offset and water level of every lake by pointer jumping ---------------------------------------------
// The connections make a forest over the lakes: a connected lake hangs below the lake it drains into, the river mouths
// are the roots.  calculateflow's recursion reaches a lake at depth (parent's depth + depth of the cell it drains into
// + 1), and the lake drawing gives it the highest pass on the way down (:683-690) -- a sum and a max along the path to
// the root, i.e. pointer doubling: ~log2(lakes) sweeps over flat arrays instead of a host walk that misses the cache at
// every lake (0.3 s for 2.6 M lakes).
__global__ void k_lake_jump(const int *__restrict__ pin, const int *__restrict__ oin, const float *__restrict__ lin,
                            int *__restrict__ pout, int *__restrict__ oout, float *__restrict__ lout, int n, int *__restrict__ changed)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pin[i], pp = pin[p];
    oout[i] = oin[i] + oin[p];                          // (a root carries 0 and points at itself)
    lout[i] = fmaxf(lin[i], lin[p]);
    pout[i] = pp;
    // a sweep in which no pointer moves changes nothing (0 is added, the maximum is taken with itself): the forest is flat
    if (changed && pp != p && *changed == 0) *changed = 1;
}
// per lake: into the per-cell tables at its sink cell, if its root is a river mouth
__global__ void k_lake_scatter(const int *__restrict__ root, const int *__restrict__ off, const float *__restrict__ lvl,
                               const unsigned char *__restrict__ is_mouth, const int *__restrict__ sink_cell,
                               int *__restrict__ lake_off, float *__restrict__ level_or_null, int *__restrict__ reached, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!is_mouth[root[i]]) return;                     // never reached from a river mouth: stays -1 / NaN
    lake_off[sink_cell[i]] = off[i];
    if (level_or_null) level_or_null[sink_cell[i]] = lvl[i];
    atomicAdd(reached, 1);
}

// ---- upstream-area accumulation by depth level --------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_gdepth(const int *__restrict__ label, const int *__restrict__ dist, const int *__restrict__ lake_off,
                                                     int *__restrict__ gdepth, int *__restrict__ maxd, size_t n)
{
    // the deepest cell: one atomic per block of FLOW_ITEMS x 256 cells (see block_append_base: one per wave took 0.63 ms at 4096x2048)
    __shared__ int s_max[4];
    int m = -1;
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const size_t i = ((size_t)blockIdx.x * FLOW_ITEMS + t) * 256 + threadIdx.x;
        if (i >= n) continue;
        const int l = label[i];
        int g = -1;
        if (l >= 0 && lake_off[l] >= 0) g = dist[i] + lake_off[l];
        gdepth[i] = g;
        m = max(m, g);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (m > 0) atomicMax(maxd, m);
    }
}

__global__ void k_flow_iota(unsigned *__restrict__ v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (unsigned)i;
}

// start[l] = first position in the depth-sorted order whose depth is >= l (l = 0..levels); unreached cells (-1) sort last
__global__ void k_flow_level_starts(const unsigned *__restrict__ sorted_depth, size_t n, int *__restrict__ start, int levels)
{
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > levels) return;
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (sorted_depth[mid] < (unsigned)l) lo = mid + 1; else hi = mid;
    }
    start[l] = (int)lo;
}

__global__ void k_flow_level(flow_canvas fc, const int *__restrict__ order, int first, int count,
                             const unsigned short *__restrict__ mask, const int *__restrict__ conn_from,
                             const float *__restrict__ own_row, float *__restrict__ acc)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    int W = fc.c.W, H = fc.c.H;
    int p = order[first + t];
    int x0 = p % W, y0 = p / W;
    float sum = own_row[y0];                            // pixelwidthy*pixelwidthx*cos(geoy)*1e-5 (:608-613)
    unsigned m = mask[p];
#pragma unroll
    for (int k = 0; k < 8; k++) {                       // sum += rec(n) in FlowFilter::neighbours order (:614-616)
        if (!(m & (unsigned)c_nbit[k])) continue;
        int x = x0 + c_ndx[k], y = y0 + c_ndy[k];
        if (fc.wrap) x = (x + W) % W; else if (x < 0 || x >= W) continue;
        if (y < 0 || y >= H) continue;
        sum = __fadd_rn(sum, acc[y * W + x]);
    }
    int from = conn_from[p];
    if (from >= 0) sum = __fadd_rn(sum, acc[from]);     // sum += rec(connections[p].from) (:617-619)
    acc[p] = sum;
}

// Upstream-area sums as a dataflow instead of one launch per depth level.  The level-synchronous form is bound by the
// ~5 us a dependent launch costs, times the depth of the drainage forest (1 800 levels at 4096x2048: 9 ms of a 21 ms
// erosion round); the arithmetic itself is a few adds per cell.  Here a persistent grid walks the cells in depth order,
// deepest first, 64 cells per wave ticket; a cell waits until the sums it adds have been published (a sum is its own
// ready flag: the array starts as NaN bit patterns), adds them in FlowFilter::neighbours order plus its connected lake
// -- the reference's fp32 sums, whatever the timing -- and publishes its own.  Every cell a ticket waits for has a
// smaller ticket, held by a wave that is already running, so the grid always drains; the critical path is the longest
// river times one L2 round trip.  Values cross CUs through sc1 (L2) loads and stores.  A poll budget turns a protocol
// error into a failed call instead of a hung GPU.
#define FLOW_NOT_READY 0xFFFFFFFFu
__device__ __forceinline__ float dpp_prev_lane(float v)      // lane l <- lane l - 1 (wave_shr:1; lane 0 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// what a cell adds, in the order it adds it: the upstream neighbours in FlowFilter::neighbours order (:614-616), then the lake
// connected to it (:617-619)
__device__ __forceinline__ int flow_sources(const flow_canvas &fc, int p, const unsigned short *__restrict__ mask, const int *__restrict__ conn_from, int src[9])
{
    const int W = fc.c.W, H = fc.c.H;
    const int x0 = p % W, y0 = p / W;
    const unsigned m = mask[p];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (!(m & (unsigned)c_nbit[k])) continue;
        int x = x0 + c_ndx[k], y = y0 + c_ndy[k];
        if (fc.wrap) x = (x + W) % W; else if (x < 0 || x >= W) continue;
        if (y < 0 || y >= H) continue;
        src[n++] = y * W + x;
    }
    const int from = conn_from ? conn_from[p] : -1;          // (null: the sums inside the catchments, before the lakes are connected -- phase A below)
    if (from >= 0) src[n++] = from;
    return n;
}

// Round 5, look-through: the kernel is bound by the chain of hops along the longest river -- a coherent store and a poll, ~1.1 us,
// for each of ~3000 cells (finishing the four cells in five near the sources beforehand took nothing off its time).  A cell
// that waits for exactly ONE sum need not wait for it to be PUBLISHED: while it waits it fetches what that upstream cell u itself
// adds, and as soon as all of THAT is there it forms u's sum in a register -- the same additions in the same order, so the same
// bits; u's own lane publishes it, this lane never does -- and goes on to its own.  The chain then advances two cells per hop.
// (Measured and dropped with it: a wave that takes four tickets at a time and fetches the next ticket's cells while it polls for the
// current one's -- the cells it holds without working on them are cells the front waits for: 3.2 -> 12.7 ms.  Nor does it help to
// finish the four cells in five that sit within three hops of a source in full-width passes beforehand: 3.43 -> 3.30 ms.)
__global__ __launch_bounds__(256) void k_flow_dataflow(flow_canvas fc, const int *__restrict__ order, int n_reached,
                                                       const unsigned short *__restrict__ mask, const int *__restrict__ conn_from,
                                                       const float *__restrict__ own_row, unsigned *acc, int *ticket, int *error, int look_through,
                                                       const int *__restrict__ n_dev = nullptr)
{
    const int W = fc.c.W;
    const int lane = threadIdx.x & 63;
    if (n_dev) n_reached = *n_dev;                           // (a count the host has not read back: phase A is queued before the host knows it)
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(ticket, 64);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n_reached) return;
        if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // somebody gave up: so does everybody
        const int idx = n_reached - 1 - (base + lane);       // deepest cells first
        bool pending = idx >= 0;
        int p = 0, nsrc = 0, src[9];
        float own = 0.0f;
        if (pending) {
            p = order[idx];
            own = own_row[p / W];                            // pixelwidthy*pixelwidthx*cos(geoy)*1e-5 (:608-613)
            nsrc = flow_sources(fc, p, mask, conn_from, src);
        }
        int u = -1, unsrc = 0, usrc[9];                      // the upstream cell looked through, and what IT adds
        float uown = 0.0f;
        int polls = 0;
        const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();        // (100 MHz: the budget below is wall time, whatever a poll costs)
        unsigned v[9];                                       // (a sum that has arrived is not asked for again)
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = FLOW_NOT_READY;
        while (__builtin_amdgcn_ballot_w64(pending)) {
            if (pending) {
                int missing = 0, which = 0;
                // (every sum still missing is asked for before the first answer is looked at: one round trip a poll, not one a sum)
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (k < nsrc && v[k] == FLOW_NOT_READY) v[k] = __hip_atomic_load(acc + src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if (k < nsrc && v[k] == FLOW_NOT_READY) { missing++; which = k; }
                if (missing == 1 && look_through) {
                    int cell = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) if (k == which) cell = src[k];
                    if (cell != u) {                         // (once per cell looked through)
                        u = cell;
                        uown = own_row[u / W];
                        unsrc = flow_sources(fc, u, mask, conn_from, usrc);
                    }
                    bool uready = true;
                    float usum = uown;
                    unsigned uv[9];
#pragma unroll
                    for (int k = 0; k < 9; k++)
                        if (k < unsrc) uv[k] = __hip_atomic_load(acc + usrc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < 9; k++)
                        if (k < unsrc) {
                            uready = uready && uv[k] != FLOW_NOT_READY;
                            usum = __fadd_rn(usum, __uint_as_float(uv[k]));
                        }
                    if (uready) {
#pragma unroll
                        for (int k = 0; k < 9; k++) if (k == which) v[k] = __float_as_uint(usum);
                        missing = 0;
                    }
                }
                if (missing == 0) {
                    float sum = own;
#pragma unroll
                    for (int k = 0; k < 9; k++)
                        if (k < nsrc) sum = __fadd_rn(sum, __uint_as_float(v[k]));
                    __hip_atomic_store(acc + p, __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pending = false;
                } else if ((++polls & 1023) == 0 &&          // two seconds without the sums: something upstream was never published
                           (__builtin_amdgcn_s_memrealtime() - t_wait0 > 200000000ull || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    atomicOr(error, 1);                      // (the whole grid then drains within a millisecond)
                    pending = false;
                } else __builtin_amdgcn_s_sleep(2);
            }
        }
    }
}

// ---- the sums in two phases (round 6) ---------------------------------------------------------------------------------------------
// A cell's sum is its area, its upstream neighbours' sums and -- where a lake drains into it -- that lake's total.  Which lake drains
// where is the host's Prim growth: 1.2 ms of a config-2 iteration with the GPU idle, and after it the sums waited for the hops of the
// longest river THROUGH the lakes (1 830 levels at 4096x2048).  But a lake's total only reaches the cells between the cell it enters and
// that catchment's sink: a fifteenth of the cells.  Phase A, queued before the host starts the growth and running beside it: every cell's
// sum inside its own catchment, no lake connected -- cells in order of their distance to their own sink, a few hundred levels.  Phase B,
// once the connections are known: the cells on the way from a connection's cell to its sink are marked, their sums withdrawn, and only
// those are formed again, in order of their global depth, from sums that are final (phase A's, of cells nothing drains into) or formed
// earlier in phase B.  Every sum is formed by the same additions in the same order as in one pass: the same bits.
// Phase A is not a dataflow: catchments do not talk to each other before the lakes are connected, so ONE WAVE takes ONE catchment and
// walks its cells level by level, deepest first -- no polling, no ticket, no other wave to wait for; the sums it needs it has stored
// itself, a step earlier.  (As a dataflow over all cells in order of their local depth phase A took as long as the whole one-pass form,
// 2.8 ms at 4096x2048: the grid's waves spend their time polling each other, 11 us per level.)
// keys of phase A: (dense lake id << 10 | 1023 - distance to the lake's sink) for the cells of a catchment, all ones for the rest;
// *count = how many cells have one, *maxd = the largest distance
#define FLOW_LOCAL_BITS 10
__global__ __launch_bounds__(256) void k_flow_local_keys(const int *__restrict__ label, const int *__restrict__ dense, const int *__restrict__ dist,
                                                         unsigned *__restrict__ key, unsigned *__restrict__ idx, int *count, int *maxd, size_t n)
{
    const size_t i0 = (size_t)blockIdx.x * FLOW_ITEMS * 256 + threadIdx.x;
    int c = 0, m = 0;
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const size_t i = i0 + (size_t)t * 256;
        if (i >= n) continue;
        const int l = label[i];
        unsigned k = 0xFFFFFFFFu;
        if (l >= 0) {
            const int d = dist[i];
            k = ((unsigned)dense[l] << FLOW_LOCAL_BITS) | (unsigned)((1 << FLOW_LOCAL_BITS) - 1 - min(d, (1 << FLOW_LOCAL_BITS) - 1));
            c++; m = max(m, d);
        }
        key[i] = k;
        idx[i] = (unsigned)i;
    }
    __shared__ int s_c[4], s_m[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); m = max(m, __shfl_xor(m, o)); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = c; s_m[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tc = s_c[0] + s_c[1] + s_c[2] + s_c[3], tm = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (tc) atomicAdd(count, tc);
        if (tm) atomicMax(maxd, tm);
    }
}
// off[l] = where lake l's cells start in the sorted list (every lake has its sink cell at least); off[nlakes] = the cells with a catchment
// ... and smask[i] = the receiver bits of the i-th cell of the sorted list
__global__ void k_flow_lake_offsets(const unsigned *__restrict__ skey, const int *__restrict__ order, const unsigned short *__restrict__ mask,
                                    const int *__restrict__ n_dev, int *__restrict__ off, unsigned short *__restrict__ smask, int nlakes, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nl = *n_dev;
    if (i == 0) off[nlakes] = nl;
    if (i >= (size_t)nl) return;
    smask[i] = mask[order[i]];
    const unsigned l = skey[i] >> FLOW_LOCAL_BITS;
    if (i == 0 || (skey[i - 1] >> FLOW_LOCAL_BITS) != l) off[l] = (int)i;
}
// one wave per catchment (BLOCK = 0), or one workgroup of 1024 threads for a catchment of more than FLOW_BIG_LAKE cells (BLOCK = 1: the
// largest catchment decides how long phase A takes -- 250 levels of 800 cells on one wave were 9 ms): its cells in order of their distance
// to the sink, farthest first, a level at a time.  A sum is only ever added by a cell of the same catchment -- by the same wave or the same
// workgroup, on one CU -- so the loads and stores are of wavefront / workgroup scope and stay in that CU's caches (at agent scope every one
// of them went out to the memory side of the eight L2s: 4 us a level); between two levels the wave only has to wait for its own stores
// (and the workgroup for each other's: a barrier).  The kernel's end publishes the sums to the kernels after it.
#define FLOW_BIG_LAKE 1024
#define FLOW_WINDOW 2048
// what a cell of phase A adds: its area and its upstream neighbours' sums, in FlowFilter::neighbours order (flow_sources without a lake)
// the q-th neighbour in FlowFilter::neighbours order if it drains into (x0, y0) -- bit q of the cell's receiver mask -- and is on the canvas, or -1
__device__ __forceinline__ int flow_source_at(const flow_canvas &fc, int x0, int y0, unsigned m, int q)
{
    const int W = fc.c.W, H = fc.c.H;
    int x = x0 + c_ndx[q], y = y0 + c_ndy[q];
    bool ok = (m & (unsigned)c_nbit[q]) != 0 && y >= 0 && y < H;
    if (fc.wrap) x = x < 0 ? x + W : (x >= W ? x - W : x); else ok = ok && x >= 0 && x < W;
    return ok ? y * W + x : -1;
}
template <int SCOPE>
__device__ __forceinline__ float flow_catchment_sum(const flow_canvas &fc, int p, unsigned m, const float *__restrict__ own_row, const unsigned *acc)
{
    const int W = fc.c.W, x0 = p % W, y0 = p / W;
    float sum = own_row[y0];
    // all the sums are asked for before the first is added: one round trip a level, not one a neighbour.  (Eight named values, not an
    // array: the compiler makes a vector of an array and waits for every load to insert it.)  Where they are: a cell off the canvas' edges
    // -- all but one in a thousand -- has its q-th neighbour at a fixed distance in memory (a bit test and an add); the general rule
    // (wrap-around, the rows at the poles: ~45 instructions a neighbour, and the large catchments' kernel is bound by its instructions)
    // is for the rest.
    int a0, a1, a2, a3, a4, a5, a6, a7;
    if (x0 > 0 && x0 < W - 1 && y0 > 0 && y0 < fc.c.H - 1) {
#define FLOW_NEAR(q, dx, dy, bit) a##q = (m & bit##u) ? p + (dy) * W + (dx) : -1;
        FLOW_NEAR(0, -1, -1, 1) FLOW_NEAR(1, 0, -1, 2) FLOW_NEAR(2, 1, -1, 4) FLOW_NEAR(3, -1, 0, 8)
        FLOW_NEAR(4, 1, 0, 32) FLOW_NEAR(5, -1, 1, 64) FLOW_NEAR(6, 0, 1, 128) FLOW_NEAR(7, 1, 1, 256)      // (c_ndx, c_ndy, c_nbit)
#undef FLOW_NEAR
    } else {
        a0 = flow_source_at(fc, x0, y0, m, 0); a1 = flow_source_at(fc, x0, y0, m, 1); a2 = flow_source_at(fc, x0, y0, m, 2); a3 = flow_source_at(fc, x0, y0, m, 3);
        a4 = flow_source_at(fc, x0, y0, m, 4); a5 = flow_source_at(fc, x0, y0, m, 5); a6 = flow_source_at(fc, x0, y0, m, 6); a7 = flow_source_at(fc, x0, y0, m, 7);
    }
#define FLOW_GET(q) unsigned v##q = 0; if (a##q >= 0) v##q = __hip_atomic_load(acc + a##q, __ATOMIC_RELAXED, SCOPE);
#define FLOW_ADD(q) if (a##q >= 0) sum = __fadd_rn(sum, __uint_as_float(v##q));
    FLOW_GET(0) FLOW_GET(1) FLOW_GET(2) FLOW_GET(3) FLOW_GET(4) FLOW_GET(5) FLOW_GET(6) FLOW_GET(7)
    FLOW_ADD(0) FLOW_ADD(1) FLOW_ADD(2) FLOW_ADD(3) FLOW_ADD(4) FLOW_ADD(5) FLOW_ADD(6) FLOW_ADD(7)
#undef FLOW_GET
#undef FLOW_ADD
    return sum;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK ? 1024 : 256) void k_flow_catchment_sums(flow_canvas fc, const unsigned *__restrict__ skey, const int *__restrict__ order,
                                                                          const int *__restrict__ off, const unsigned short *__restrict__ smask,
                                                                          const float *__restrict__ own_row, unsigned *acc, const int *__restrict__ list)
{
    const int lane = BLOCK ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    // (the catchments of a kind are listed -- list[1..list[0]], k_flow_lake_kinds -- and the workgroups / waves take them in turn: a grid of
    // one workgroup per lake, 43 000 of 1024 threads of which a few hundred stay, spent its time being dispatched)
    const int first = BLOCK ? (int)blockIdx.x : (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int stride = BLOCK ? (int)gridDim.x : (int)(((size_t)gridDim.x * blockDim.x) >> 6);
    for (int turn = first; turn < list[0]; turn += stride) {
    const int lake = list[1 + turn];
    int pos = off[lake];
    const int end = off[lake + 1];
    if (BLOCK) {
        __syncthreads();                                             // (the window and the counts of the catchment before)
        // The sorted list is read through a window in the LDS, 2048 entries at a time: a level then waits for the sums it adds and for its
        // own stores, not for its keys as well (they come from HBM: 4.4 us a level with them, 250 levels in the largest catchment).  The
        // last cell of a level says how many there were; one barrier a level (the count alternates between two words).
        __shared__ unsigned w_key[FLOW_WINDOW];
        __shared__ int w_ord[FLOW_WINDOW];
        __shared__ unsigned short w_msk[FLOW_WINDOW];
        __shared__ int s_cnt[2];
        int wbase = pos - FLOW_WINDOW, par = 0;                      // (nothing loaded yet)
        while (pos < end) {
            if (pos + 1024 > wbase + FLOW_WINDOW && wbase + FLOW_WINDOW < end) {
                wbase = pos;                                         // (everybody has passed the barrier of the last level read from the old window)
#pragma unroll
                for (int t = 0; t < FLOW_WINDOW / 1024; t++) {
                    const int i = wbase + t * 1024 + lane;
                    if (i < end) { w_key[t * 1024 + lane] = skey[i]; w_ord[t * 1024 + lane] = order[i]; w_msk[t * 1024 + lane] = smask[i]; }
                }
                __syncthreads();
            }
            const int w = pos - wbase + lane;
            const unsigned k0 = w_key[pos - wbase];
            const bool mine = pos + lane < end && w_key[w] == k0;    // (the cells of the level the first cell is on; sorted: a prefix of the batch)
            if (mine) {
                const int p = w_ord[w];
                if (lane == 1023 || pos + lane + 1 >= end || w_key[w + 1] != k0) s_cnt[par] = lane + 1;
                const float sum = flow_catchment_sum<__HIP_MEMORY_SCOPE_WORKGROUP>(fc, p, (unsigned)w_msk[w], own_row, acc);
                __hip_atomic_store(acc + p, __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's sums of the level have reached the L2
            __syncthreads();
            pos += s_cnt[par];
            par ^= 1;
        }
        continue;
    }
    while (pos < end) {
        // key, cell and the cell's receiver bits are asked for together (smask: the mask in the sorted order, k_flow_lake_offsets): a level
        // then waits for three round trips -- these, the sums, its own stores -- not five
        const bool have = pos + lane < end;
        const unsigned k = have ? skey[pos + lane] : 0u;
        const int p = have ? order[pos + lane] : 0;
        const unsigned m = have ? (unsigned)smask[pos + lane] : 0u;
        const unsigned k0 = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        const bool mine = have && k == k0;
        if (mine) {
            const float sum = flow_catchment_sum<__HIP_MEMORY_SCOPE_WAVEFRONT>(fc, p, m, own_row, acc);
            __hip_atomic_store(acc + p, __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pos += __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
    }
    }
}
// ... and the smallest catchments, up to FLOW_TINY_LAKE cells -- where every other cell is a lake of its own there are millions of them, and
// a wave for each is a wave for a cell or two -- one LANE each: the lane goes through its catchment's cells in the list's order, farthest
// from the sink first; what a cell adds its own lane stored before it
#define FLOW_TINY_LAKE 8
__global__ __launch_bounds__(256) void k_flow_tiny_catchment_sums(flow_canvas fc, const int *__restrict__ order, const int *__restrict__ off,
                                                                  const unsigned short *__restrict__ smask, const float *__restrict__ own_row, unsigned *acc, int nlakes)
{
    const int lake = (int)((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (lake >= nlakes) return;
    int pos = off[lake];
    const int end = off[lake + 1];
    if (end - pos > FLOW_TINY_LAKE) return;
    for (; pos < end; pos++) {
        const int p = order[pos];
        const float sum = flow_catchment_sum<__HIP_MEMORY_SCOPE_WAVEFRONT>(fc, p, (unsigned)smask[pos], own_row, acc);
        __hip_atomic_store(acc + p, __float_as_uint(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// the catchments by kind: big[0] of more than FLOW_BIG_LAKE cells, big[1..]; mid[0] of more than FLOW_TINY_LAKE, mid[1..] (a wave counts
// its lakes and takes its places in a list with one atomic)
__global__ __launch_bounds__(256) void k_flow_lake_kinds(const int *__restrict__ off, int nlakes, int *big, int *mid)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int size = l < nlakes ? off[l + 1] - off[l] : 0;
#pragma unroll
    for (int kind = 0; kind < 2; kind++) {
        const bool is = kind ? (size > FLOW_TINY_LAKE && size <= FLOW_BIG_LAKE) : size > FLOW_BIG_LAKE;
        const unsigned long long who = __builtin_amdgcn_ballot_w64(is);
        if (!who) continue;
        int *list = kind ? mid : big;
        int base = 0;
        if (lane == 0) base = atomicAdd(list, __builtin_popcountll(who));
        base = __builtin_amdgcn_readfirstlane(base);
        if (is) list[1 + base + __builtin_popcountll(who & ((1ull << lane) - 1ull))] = l;
    }
}

// the cell a cell drains to (k_flow_parent's rule), or -1
__device__ __forceinline__ int flow_receiver_of(const flow_canvas &fc, int cd, int i)
{
    const int W = fc.c.W, H = fc.c.H;
    if (cd == 0 || cd == 5) return -1;
    const int dx = (cd - 1) % 3 - 1, dy = (cd - 1) / 3 - 1;
    int x = i % W + dx, y = i / W + dy;
    if (fc.wrap) { if (x < 0) x += W; else if (x >= W) x -= W; } else if (x < 0 || x >= W) return -1;
    if (y < 0 || y >= H) return -1;
    return y * W + x;
}

// phase B, marking: from every cell a lake drains into, downstream to the catchment's sink (or to a cell somebody else has marked: the
// one who marked it goes on from there)
// ... and where every marked cell's sum goes next: the cell it drains to, or -- for a lake's sink -- the cell the lake drains into (next[],
// -1 everywhere before; a river mouth keeps -1)
__global__ __launch_bounds__(256) void k_flow_mark_affected(flow_canvas fc, const unsigned char *__restrict__ code, const int *__restrict__ conn_from,
                                                            const int *__restrict__ gdepth, int *__restrict__ marked, int *__restrict__ next, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || conn_from[i] < 0 || gdepth[i] < 0) return;
    next[conn_from[i]] = (int)i;
    // (a step is one round trip: where the cell drains to is known from its code, fetched a step ahead, so the next cell's code is on its
    // way while the mark is being set)
    int q = (int)i, cd = code[q];
    for (int guard = 0; guard < (1 << 24); guard++) {
        const int r = flow_receiver_of(fc, cd, q);
        const int ncd = r >= 0 ? (int)code[r] : 0;
        if (atomicExch(marked + q, 1) != 0) break;
        if (r < 0) break;
        next[q] = r;
        q = r; cd = ncd;
    }
}

// phase B, the list: (global depth, cell) of every marked cell, its sum withdrawn
__global__ __launch_bounds__(256) void k_flow_collect_affected(const int *__restrict__ marked, const int *__restrict__ gdepth, unsigned *acc,
                                                               unsigned *__restrict__ key, unsigned *__restrict__ cell, int *count, size_t n)
{
    int mine[FLOW_ITEMS];
    int c = 0;
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        const size_t i = ((size_t)blockIdx.x * FLOW_ITEMS + t) * 256 + threadIdx.x;
        mine[t] = (i < n && marked[i] && gdepth[i] >= 0) ? (int)i : -1;
        c += mine[t] >= 0;
    }
    int slot = block_append_base(c, count);
#pragma unroll
    for (int t = 0; t < FLOW_ITEMS; t++) {
        if (mine[t] < 0) continue;
        acc[mine[t]] = FLOW_NOT_READY;
        key[slot] = (unsigned)gdepth[mine[t]];
        cell[slot] = (unsigned)mine[t];
        slot++;
    }
}

// ---- phase B as a walk along the rivers (round 6) ---------------------------------------------------------------------------------------
// As a dataflow over its cells in depth order phase B is one coherent store and one poll per cell of the longest river: ~0.57 us x 1 830
// levels = 1.05 ms of a config-2 iteration, for 35 000 cells.  Here the marked cells are cut into REACHES -- stretches of river, each cell
// draining into the next, through the lakes' sinks into the cells the lakes drain into (k_reach_choose) -- and every other sum a cell of a
// reach adds is either final (phase A's) or the sum of the LAST cell of another reach, one level deeper.  So the cells are laid out reach by reach, the reaches in order of the depth of their last
// cell, deepest first: whatever a cell waits for lies BEFORE it in the list.  A wave takes 64 consecutive cells; every lane collects the
// sums its cell adds from outside its reach (polling, as in the dataflow) and the wave then runs down the reaches in registers -- lane l
// takes lane l-1's sum through a DPP shift, adds it where the reference adds it (between the neighbours before it and those after it,
// FlowFilter::neighbours order, then the connected lake: the same additions in the same order, the same bits) and hands its own on: a
// few cycles per cell instead of a round trip to the L2.  Only a reach's first lane in a wave waits for another wave.  A sum not there yet
// is the NaN it is stored as: it runs down its reach as NaN, and what is NaN after the run is not published and tried again.
// The reaches.  Where several marked cells drain into one, the reach goes on from the one with the LONGEST marked river above it (any choice
// gives the same sums; this one keeps the longest river -- the one everything waits for -- in as few reaches as its confluences allow, and
// a reach that ends costs the next one a round trip to the L2).  top[c] = the greatest global depth among the marked cells above c, c
// included: pointer jumping down next[], synchronous rounds (two copies of top and of the pointers), log2(longest river) of them.
// (what another thread of the same launch may have written is read with a coherent load)
__device__ __forceinline__ int reach_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct reach_tables {
    const unsigned *cell;            // the marked cells, n of them
    const int *gdepth, *next, *marked;
    int *top[2], *hop[2], *tail;
    unsigned long long *key;
    int n, rounds, depth_bits, cell_bits, max_depth;
};
__device__ __forceinline__ void reach_init(const reach_tables &t, int i)
{
    const unsigned c = t.cell[i];
    t.top[0][c] = t.top[1][c] = t.gdepth[c];
    t.hop[0][c] = t.next[c];
    t.tail[c] = (int)c;                                     // (reach_choose: the reach goes on, for all but its last cell)
}
__device__ __forceinline__ void reach_height(const reach_tables &t, int i, int from)
{
    const unsigned c = t.cell[i];
    const int h = reach_peek(t.top[from] + c), v = reach_peek(t.hop[from] + c);
    atomicMax(t.top[from ^ 1] + c, h);                      // (the other copy is two rounds old)
    if (v >= 0) atomicMax(t.top[from ^ 1] + v, h);
    t.hop[from ^ 1][c] = v >= 0 ? reach_peek(t.hop[from] + v) : -1;
}
// for every marked cell: which of the marked cells draining into it the reach comes from; tail[u] = c for that one (the reach goes on), the
// others keep tail[u] = u (their reach ends)
__device__ __forceinline__ void reach_choose(const reach_tables &t, const flow_canvas &fc, const unsigned short *__restrict__ mask,
                                             const int *__restrict__ conn_from, int i, int from)
{
    const int c = (int)t.cell[i];
    int src[9];
    const int nsrc = flow_sources(fc, c, mask, conn_from, src);
    int best = -1, best_top = -1;
#pragma unroll
    for (int k = 0; k < 9; k++)
        if (k < nsrc && t.marked[src[k]]) { const int h = reach_peek(t.top[from] + src[k]); if (h > best_top) { best_top = h; best = src[k]; } }
    if (best >= 0) t.tail[best] = c;
}
// tail[c] = the last cell of c's reach (pointer jumping in place: a pointer only ever moves down its own reach)
__device__ __forceinline__ void reach_tail(const reach_tables &t, int i)
{
    const unsigned c = t.cell[i];
    const int e = reach_peek(t.tail + c);
    const int ee = reach_peek(t.tail + e);
    if (ee != e) __hip_atomic_store(t.tail + c, ee, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// keys: (depth of the reach's last cell, the reach -- its last cell --, the cell's own depth), all deepest first
__device__ __forceinline__ void reach_key(const reach_tables &t, int i)
{
    const unsigned c = t.cell[i];
    const unsigned w = (unsigned)reach_peek(t.tail + c);
    t.key[i] = ((unsigned long long)(unsigned)(t.max_depth - t.gdepth[w]) << (t.cell_bits + t.depth_bits)) | ((unsigned long long)w << t.depth_bits) |
               (unsigned long long)(unsigned)(t.max_depth - t.gdepth[c]);
}
// one launch per step (any number of cells) ...
__global__ void k_reach_init(reach_tables t) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < t.n) reach_init(t, i); }
__global__ void k_reach_height(reach_tables t, int from) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < t.n) reach_height(t, i, from); }
__global__ void k_reach_choose(reach_tables t, flow_canvas fc, const unsigned short *__restrict__ mask, const int *__restrict__ conn_from, int from)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.n) reach_choose(t, fc, mask, conn_from, i, from);
}
__global__ void k_reach_tail(reach_tables t) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < t.n) reach_tail(t, i); }
__global__ void k_flow_walk_keys(reach_tables t) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < t.n) reach_key(t, i); }
// (Measured and dropped: ONE workgroup of 1024 threads taking all the steps with a barrier between them instead of 2 x log2(levels) + 3
// launches -- 35 000 cells are 34 per thread, each a chain of coherent loads and an atomic, one after the other: 1.5 ms against the
// launches' 0.17 ms.)

__device__ __forceinline__ bool flow_is_nan(float x) { return (__float_as_uint(x) & 0x7FFFFFFFu) > 0x7F800000u; }

__global__ __launch_bounds__(256) void k_flow_walk(flow_canvas fc, const unsigned long long *__restrict__ skey, const int *__restrict__ cells, int n, int depth_bits,
                                                   const unsigned short *__restrict__ mask, const int *__restrict__ conn_from,
                                                   const float *__restrict__ own_row, unsigned *acc, int *ticket, int *error)
{
    const int W = fc.c.W;
    const int lane = threadIdx.x & 63;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(ticket, 64);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) return;
        if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const int i = base + lane;
        const bool valid = i < n;
        int p = 0, nsrc = 0, src[9], h = 9;                  // h: which of the sums the cell adds is its upstream neighbour in the reach (9: none in this wave)
        float own = 0.0f;
        bool linked = false;                                 // the lane before holds the cell before, in the same reach
        if (valid) {
            p = cells[i];
            own = own_row[p / W];
            nsrc = flow_sources(fc, p, mask, conn_from, src);
            if (lane > 0 && (skey[i] >> depth_bits) == (skey[i - 1] >> depth_bits)) {
                const int before = cells[i - 1];
#pragma unroll
                for (int k = 0; k < 9; k++) if (k < nsrc && src[k] == before) h = k;
                linked = h < 9;
                if (!linked) atomicOr(error, 2);             // (the cell before does not drain into this one: not a reach)
            }
        }
        // how far a sum has to run inside this wave: the longest stretch of linked lanes
        const unsigned long long heads = __builtin_amdgcn_ballot_w64(!linked);
        const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        int run = lane - (63 - __builtin_clzll(below));     // (lane 0 is never linked: `below` has a bit)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) run = max(run, __shfl_xor(run, o));
        const int hops = __builtin_amdgcn_readfirstlane(run);
        int nafter = linked ? nsrc - 1 - h : 0;
        int most_after = nafter;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) most_after = max(most_after, __shfl_xor(most_after, o));
        most_after = __builtin_amdgcn_readfirstlane(most_after);

        unsigned v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = FLOW_NOT_READY;
        bool done = !valid;
        float s = __uint_as_float(FLOW_NOT_READY);
        int polls = 0;
        const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_ballot_w64(!done)) {
            // the sums from outside the reach: every one still missing is asked for, then looked at
#pragma unroll
            for (int k = 0; k < 9; k++)
                if (!done && k < nsrc && k != h && v[k] == FLOW_NOT_READY) v[k] = __hip_atomic_load(acc + src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ready = true;
#pragma unroll
            for (int k = 0; k < 9; k++)
                if (k < nsrc && k != h && v[k] == FLOW_NOT_READY) ready = false;
            // before: own + the sums added before the reach's (all of them for a cell that heads its reach here); after[j]: those added after it
            float before = own, after[8];
#pragma unroll
            for (int k = 0; k < 9; k++)
                if (k < nsrc && k < h) before = __fadd_rn(before, __uint_as_float(v[k]));
            if (!ready) before = __uint_as_float(FLOW_NOT_READY);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                after[j] = 0.0f;
#pragma unroll
                for (int k = 1; k < 9; k++) if (k == h + 1 + j) after[j] = __uint_as_float(v[k]);
            }
            float cur = done ? s : (linked ? __uint_as_float(FLOW_NOT_READY) : before);
            for (int hop = 0; hop < hops; hop++) {
                const float carried = dpp_prev_lane(cur);
                float t = __fadd_rn(before, carried);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (j < most_after) { if (j < nafter) t = __fadd_rn(t, after[j]); }
                if (linked && !done) cur = t;
            }
            const bool publish = !done && !flow_is_nan(cur);
            if (publish) {
                __hip_atomic_store(acc + p, __float_as_uint(cur), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s = cur;
                done = true;
            }
            if (__builtin_amdgcn_ballot_w64(publish) == 0) {
                if ((++polls & 1023) == 0 &&                 // two seconds without a sum: something upstream was never published
                    (__builtin_amdgcn_s_memrealtime() - t_wait0 > 200000000ull || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    atomicOr(error, 1);
                    done = true;
                } else __builtin_amdgcn_s_sleep(1);
            }
        }
    }
}

// lakeID[p] = pow(sum, exponent) (:620) for every reached cell, one launch after the level sums.  The reference's
// unqualified pow(float, float) is the C double function under g++ (its headers pull <cmath> only), rounded to float
// by the assignment: double pow, one rounding.
__global__ void k_flow_pow(const int *__restrict__ gdepth, const float *__restrict__ acc, float *__restrict__ out, float exponent, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || gdepth[i] < 0) return;
    out[i] = (float)pow((double)acc[i], (double)exponent);
}

// lake drawing (:660-699): cells of a reached lake's catchment at or below the lake's water level become 0
__global__ void k_flow_lakefill(const float *__restrict__ h, const int *__restrict__ label, const float *__restrict__ level,
                                float *__restrict__ flow, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = label[i];
    if (l < 0) return;
    float wl = level[l];                                // NaN for lakes the walk from the river mouths never reaches
    if (h[i] <= wl) flow[i] = 0.0f;
}

__global__ void k_flow_scatter_f(const int *__restrict__ idx, const float *__restrict__ v, float *__restrict__ dst, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = v[i];
}

__global__ void k_fill_f(float *p, float v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}


// ---- the growth of the lake graph in parallel (round 3) -------------------------------------------------------------
// solvingConnections (FlowFilter.cpp:533-595) grows a forest from the river mouths by always taking the lowest pass
// (h, from, s) on the frontier: Prim's algorithm on a directed graph, 0.45 s of one host thread at 16384x8192.  Its
// structure (measured: 2.4 M connections, the deepest lake ~1100 hops from its mouth, 1.5 M "cascades" of which the largest
// holds 21 000 lakes): whenever the pass it takes is higher than every pass before, a CASCADE starts -- the lake behind
// that pass and everything reachable from it over lower passes are placed before any higher pass is looked at.  The level
// at which a lake is placed is its minimax distance d from the mouths (the lowest possible highest pass on a way to it):
// the lakes of equal d are one cascade, cascades never compete for a lake, they follow each other in order of d, and inside
// a cascade the growth is Prim from its first lake over the cascade's own lakes.  d is a fixed point that does not care
// about order, d(F) = min over passes (L -> F) of max(d(L), rank of the pass): computed here on the device by frontier
// relaxation with atomicMin; the cascades then run on all host cores, and the connections come out in the very order the
// one-thread growth makes them (cascades by d, each in its own pop order).  Same forest, same bits (tests/test_flow.py).
__device__ __forceinline__ unsigned flow_hkey(int hbits)
{
    float h = __int_as_float(hbits);
    if (h == 0.0f) h = 0.0f;                                  // -0.0 == +0.0 in the growth's comparison
    const unsigned u = __float_as_uint(h);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone in h
}
__global__ void k_growth_keys(const int4 *__restrict__ ent, unsigned *__restrict__ ks, int *__restrict__ idx, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { ks[i] = (unsigned)ent[i].z; idx[i] = i; }
}
__global__ void k_growth_keys2(const int4 *__restrict__ ent, const int *__restrict__ idx, unsigned long long *__restrict__ khf, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { const int4 e = ent[idx[j]]; khf[j] = ((unsigned long long)flow_hkey(e.x) << 32) | (unsigned)e.y; }
}
__global__ void k_growth_rank(const int *__restrict__ order, unsigned *__restrict__ rank, int n)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) rank[order[r]] = (unsigned)r + 1u;             // ranks 1 .. n; 0 = a river mouth's level
}
__global__ void k_growth_init(unsigned *__restrict__ d, int *__restrict__ queued, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { d[i] = 0xFFFFFFFFu; queued[i] = 0; }
}
__global__ void k_growth_seed(const int *__restrict__ mouths, int nm, unsigned *__restrict__ d, int *__restrict__ frontier, int *__restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nm) { d[mouths[i]] = 0u; frontier[i] = mouths[i]; }
    if (i == 0) { count[0] = nm; count[1] = 0; count[2] = 0; }
}
// one relaxation round: every lake of the frontier offers max(d(L), rank) to the lakes behind its passes.  A lake whose d
// falls joins the next round's frontier once: `stamp` holds the last round a lake was queued for, and only the thread that
// raises it to round + 1 appends the lake.  No fences: a lake that reads a stale d of its own in this round has been (or
// will be) queued for the next one by whoever lowered it, and the next round is another launch.
// (Measured and dropped: all rounds in one launch with a grid-wide barrier -- 116 ms instead of 68 at 16384x8192, 17 instead
// of 9 at 4096x2048: a round is a chain of dependent memory round trips, not launch overhead, and the persistent grid has
// fewer threads to hide them behind.)
__global__ __launch_bounds__(256) void k_growth_relax(const int *__restrict__ fin, const int *__restrict__ nin, int *__restrict__ fout, int *__restrict__ nout,
                                                       unsigned *__restrict__ d, const int2 *__restrict__ span, const int *__restrict__ vstart,
                                                       const int4 *__restrict__ ent, const unsigned *__restrict__ rank, int *__restrict__ stamp, int round,
                                                       int *__restrict__ nclear)
{
    // (three counters in turn: this round reads one, fills the next and clears the third -- the one the round before read and the
    // round after fills -- so no launch stands between two rounds)
    if (blockIdx.x == 0 && threadIdx.x == 0) *nclear = 0;
    // the next frontier is collected per block in LDS and appended with ONE add on the global counter (block_append_base's
    // reason: an add per lake on one address costs ~5 ns per wave-instruction, serialised; the order inside a frontier
    // does not matter, the levels are a fixed point)
    constexpr int QCAP = 2048;
    __shared__ int q[QCAP];
    __shared__ int qn, qbase;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const int n = *nin;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int L = fin[t];
        const unsigned dl = __hip_atomic_load(&d[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int e1 = span[L].y;
        for (int i = vstart[L]; i < e1; i++) {
            const int F = ent[i].y;
            const unsigned nd = max(dl, rank[i]);
            if (nd < __hip_atomic_load(&d[F], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const unsigned old = atomicMin(&d[F], nd);
                if (nd < old && atomicMax(&stamp[F], round + 1) < round + 1) {
                    const int k = atomicAdd(&qn, 1);
                    if (k < QCAP) q[k] = F;
                    else fout[atomicAdd(nout, 1)] = F;      // (the block's queue is full: straight to the list)
                }
            }
        }
    }
    __syncthreads();
    const int m = min(qn, QCAP);
    if (threadIdx.x == 0 && m > 0) qbase = atomicAdd(nout, m);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) fout[qbase + i] = q[i];
}
__global__ void k_growth_iota(int *__restrict__ v, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = i; }
// the pass that opens the cascade of every lake in sorted order (only read at a cascade's first lake)
__global__ void k_growth_heads(const unsigned *__restrict__ dsorted, const int *__restrict__ order, int *__restrict__ head, int n, int nent)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned x = dsorted[p];
    head[p] = (x >= 1u && x <= (unsigned)nent) ? order[x - 1] : -1;
}

// ---- host ---------------------------------------------------------------------------------------------------------
namespace {
struct Pass { float h; int from; int tolocation; };
struct PassLess {
    bool operator()(const Pass &a, const Pass &b) const
    { return std::tie(a.h, a.from, a.tolocation) < std::tie(b.h, b.from, b.tolocation); }
};
}  // namespace

// Device-resident flow accumulation.  d_h: heights (W*H); d_sel: selection mask or nullptr; d_flow: result.
// allocator for big host tables that are visited in random order: 2 MB-aligned and advised as huge pages
template <class T> struct thp_allocator {
    typedef T value_type;
    thp_allocator() = default;
    template <class U> thp_allocator(const thp_allocator<U> &) {}
    T *allocate(size_t n)
    {
        void *p = nullptr;
        const size_t bytes = std::max<size_t>(n * sizeof(T), 1);
        if (bytes < ((size_t)32 << 20)) {                              // small tables: the heap's recycled pages beat fresh huge ones
            if (!(p = malloc(bytes))) throw std::bad_alloc();
            return (T *)p;
        }
        if (posix_memalign(&p, (size_t)2 << 20, bytes)) throw std::bad_alloc();
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        return (T *)p;
    }
    void deallocate(T *p, size_t) { free(p); }
    template <class U> bool operator==(const thp_allocator<U> &) const { return true; }
    template <class U> bool operator!=(const thp_allocator<U> &) const { return false; }
};

int flow_device(dmg_ctx *ctx, const dmg_grid &grid, const dmg_topology &topo, const float *d_h, const float *d_sel,
                float preblur, float exponent, float *d_flow, int lakes)
{
    const int W = grid.W, H = grid.H;
    const size_t N = (size_t)W * H;
    hipStream_t st = ctx->stream;
    const unsigned nb = (unsigned)((N + 255) / 256);
    static const bool timing = dmg_hook("DMG_FLOW_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[flow] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };

    flow_canvas fc;
    fc.c = make_canvas(grid, topo);
    dmg_grid gm = grid;                                  // coordsMod (:253-256): float += double literal
    gm.coords[0] = (float)((double)gm.coords[0] + 1e-3);
    gm.coords[1] = (float)((double)gm.coords[1] - 1e-3);
    fc.cm = make_canvas(gm, dmg_topology_of(gm));
    fc.wrap = fabs(grid.coords[3] - grid.coords[2]) > 2 * M_PI - 1e-4;       // :47

    // per-row tables: pixelsize() of the direction pass, cell area of calculateflow (double arithmetic, :608-613); kept with the context
    // from one solve of a grid to the next (dmg_ctx::flow_rows)
    if (!ctx->flow_rows || memcmp(&ctx->flow_rows_grid, &grid, sizeof(dmg_grid)) != 0) {
        std::vector<float> rows((size_t)3 * H);
        float pwx = (float)(grid.circumference * (grid.coords[3] - grid.coords[2]) / (2 * M_PI) / W);
        float pwy = (float)(grid.circumference * (grid.coords[1] - grid.coords[0]) / (2 * M_PI) / H);
        for (int y = 0; y < H; y++) {
            dmg_row_pixelsize(gm, y, &rows[y], &rows[(size_t)H + y]);
            float fy = (float)y / H;
            float geoy = (fy * (grid.coords[1] - grid.coords[0]) + grid.coords[0]);
            rows[(size_t)2 * H + y] = (float)(pwy * pwx * cos((double)geoy) * 1e-5);      // unqualified cos(float) = the C double cos there
        }
        DMG_HIP(hipStreamSynchronize(st));                   // (a solve still using the old tables)
        if (ctx->flow_rows) { (void)hipFree(ctx->flow_rows); ctx->flow_rows = nullptr; }
        if (hipMalloc((void **)&ctx->flow_rows, rows.size() * 4) != hipSuccess) { ctx->flow_rows = nullptr; return dmg_fail(DMG_ENOMEM, "hipMalloc(%zu) failed", rows.size() * 4); }
        DMG_HIP(hipMemcpyAsync(ctx->flow_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        DMG_HIP(hipStreamSynchronize(st));                   // `rows` is a host temporary
        ctx->flow_rows_grid = grid;
    }
    struct row_table { const float *p; const float *f() const { return p; } };
    const row_table ddx{ ctx->flow_rows }, ddy{ ctx->flow_rows + H }, down{ ctx->flow_rows + (size_t)2 * H };
    dev_buf hb, tmp, code, mask, par[2], dst[2], label, changed, count;
    int rc;
    DMG_ALLOC(hb, N * 4); DMG_ALLOC(tmp, N * 4); DMG_ALLOC(code, N); DMG_ALLOC(mask, N * 2);
    DMG_ALLOC(par[0], N * 4); DMG_ALLOC(par[1], N * 4); DMG_ALLOC(dst[0], N * 4); DMG_ALLOC(dst[1], N * 4);
    DMG_ALLOC(label, N * 4); DMG_ALLOC(changed, 128 * 4); DMG_ALLOC(count, 4);      // changed: a word per round (cleared once, not once per round)

    // Without sea there is no river mouth, and without a river mouth no lake is ever reached (below): the flow is -1 everywhere.  A terrain
    // that is positive everywhere stays positive under the blur -- a tap is a chain of lerps between texels (fma(w, b - a, a) with w in
    // [0, 1) lies between a and b, rounding is monotone), a pass a sum of taps times positive weights; 1e-20 keeps the products clear of
    // underflow -- so no cell is sea (`img <= 0`, k_flow_direction / k_flow_mask) and no neighbour code is 0.  Looked for only where the
    // last solve of this size found no mouth (config 5's iteration: 9 ms of blur, 2.6 ms of direction and mask for a field of -1).
    const unsigned nbi0 = (unsigned)((N + (size_t)256 * FLOW_ITEMS - 1) / ((size_t)256 * FLOW_ITEMS));
    if (ctx->flow_hint_no_mouths && ctx->flow_hint_cells == N && !d_sel) {
        DMG_HIP(hipMemsetAsync(count.p, 0, 4, st));
        hipLaunchKernelGGL(k_flow_any_sea, dim3(nbi0), dim3(256), 0, st, d_h, count.i(), N);
        int any = 0;
        DMG_HIP(hipMemcpyAsync(&any, count.p, 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        if (!any) {
            hipLaunchKernelGGL(k_fill_f, dim3(nb), dim3(256), 0, st, d_flow, -1.0f, N);        // unreached cells keep -1 (:598-604)
            DMG_HIP(hipGetLastError());
            lap("no sea, no river mouth: -1 everywhere");
            return DMG_OK;
        }
    }
    // findMagicNumbers: pre-blur (:90-103) then the two passes on the blurred terrain
    DMG_HIP(hipMemcpyAsync(hb.p, d_h, N * 4, hipMemcpyDeviceToDevice, st));
    if (preblur > 0.0f && (rc = blur_device(ctx, grid, topo, hb.f(), tmp.f(), preblur))) return rc;
    dim3 g2((W + 255) / 256, H);
    hipLaunchKernelGGL(k_flow_direction, g2, dim3(256), 0, st, fc, hb.f(), d_sel, ddx.f(), ddy.f(), (unsigned char *)code.p);
    hipLaunchKernelGGL(k_flow_mask, g2, dim3(256), 0, st, fc, hb.f(), d_sel, (const unsigned char *)code.p, (unsigned short *)mask.p);
    DMG_HIP(hipGetLastError());

    lap("alloc+blur+direction+mask");
    // Without a river mouth no lake is ever reached (solvingConnections starts at the mouths, :533-595; calculateflow at what it connected):
    // the flow is -1 everywhere, whatever the catchments are.  Config 5's iteration meets this every time -- the terrain the tectonics
    // render pass leaves has no sink beside the sea -- and spent 7 ms of its 34 on catchments nobody reads.  Where the last solve of this
    // size found no mouth, this one looks for one before anything else (0.2 ms and a synchronisation; a grid that has mouths never pays).
    if (ctx->flow_hint_no_mouths && ctx->flow_hint_cells == N) {
        DMG_HIP(hipMemsetAsync(count.p, 0, 4, st));
        hipLaunchKernelGGL(k_flow_any_mouth, dim3((unsigned)((N + (size_t)256 * FLOW_ITEMS - 1) / ((size_t)256 * FLOW_ITEMS))), dim3(256), 0, st,
                           (const unsigned short *)mask.p, count.i(), N);
        int any = 0;
        DMG_HIP(hipMemcpyAsync(&any, count.p, 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        if (!any) {
            hipLaunchKernelGGL(k_fill_f, dim3(nb), dim3(256), 0, st, d_flow, -1.0f, N);        // unreached cells keep -1 (:598-604)
            DMG_HIP(hipGetLastError());
            lap("no river mouth: -1 everywhere");
            return DMG_OK;
        }
        ctx->flow_hint_no_mouths = false;
    }
    // catchment labels + depth by pointer jumping (log2 of the longest river iterations)
    hipLaunchKernelGGL(k_flow_parent, dim3(nb), dim3(256), 0, st, fc, (const unsigned char *)code.p, par[0].i(), dst[0].i());
    int cur = 0;
    DMG_HIP(hipMemsetAsync(changed.p, 0, 128 * 4, st));
    // (a round past the last one changes nothing: rivers of 2^8 cells and more are the rule, so the first rounds run without asking; a
    // readback costs more than the round.  From which round on to ask: the one the last solve of this size ended with -- one question
    // where the seventh round on were three -- and every 32nd solve two rounds earlier, so that the hint follows rivers that got shorter.)
    // (The hint works downwards too: the plate map the tectonics pass leaves is a sink per cell -- nothing moves in the first round, and
    // seven rounds at the HBM roof went by before the first question: 3 ms of config 5's iteration.)
    int first_ask = N >= ((size_t)1 << 20) ? 7 : 0;
    if (ctx->flow_hint_cells == N) first_ask = std::max(0, ctx->flow_hint_rounds - ((++ctx->flow_hint_age & 31u) == 0 ? 2 : 0));
    for (int it = 0; it < 40; it++) {
        hipLaunchKernelGGL(k_flow_jump, dim3(nb), dim3(256), 0, st, par[cur].i(), dst[cur].i(), par[cur ^ 1].i(), dst[cur ^ 1].i(), N, changed.i() + it, it == 0 ? 1 : 0);
        DMG_HIP(hipGetLastError());
        cur ^= 1;
        if (it < first_ask) continue;
        int ch[40];                                          // (every round's word so far: the hint is the FIRST round in which nothing moved)
        DMG_HIP(hipMemcpyAsync(ch, changed.i(), (size_t)(it + 1) * 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        int still = 0;
        while (still <= it && ch[still]) still++;
        if (still <= it) { ctx->flow_hint_cells = N; ctx->flow_hint_rounds = still; break; }
    }
    hipLaunchKernelGGL(k_flow_label, dim3(nb), dim3(256), 0, st, par[cur].i(), (const unsigned char *)code.p, label.i(), N);

    lap("pointer jumping");
    // dense lake ids
    dev_buf sflag, dense, sinkc, scan_tmp;
    DMG_ALLOC(sflag, N * 4); DMG_ALLOC(dense, N * 4 + 4);
    hipLaunchKernelGGL(k_flow_sinkflag, dim3(nb), dim3(256), 0, st, label.i(), sflag.i(), N);
    {
        size_t sb = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, sb, sflag.i(), dense.i(), (int)N, st);
        DMG_ALLOC(scan_tmp, sb + 256);
        DMG_HIP(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, sb, sflag.i(), dense.i(), (int)N, st));
    }
    // river mouths (dense ids), compacted (sorted below, on the device) -- counted with the lakes: without a mouth no pass is ever looked
    // at (below), and the pass candidates need not be found either (4.5 ms of config 5's iteration at 16384x8192, whose terrain has none)
    const unsigned nbi = (unsigned)((N + (size_t)256 * FLOW_ITEMS - 1) / ((size_t)256 * FLOW_ITEMS));     // blocks of FLOW_ITEMS x 256 cells (k_flow_mouths, k_flow_gdepth)
    dev_buf nsel, mouths_d;
    DMG_ALLOC(nsel, 8);
    DMG_ALLOC(mouths_d, N * 4);
    DMG_HIP(hipMemsetAsync(nsel.p, 0, 8, st));
    hipLaunchKernelGGL(k_flow_mouths, dim3(nbi), dim3(256), 0, st, (const unsigned short *)mask.p, dense.i(), mouths_d.i(), nsel.i(), N);
    int nlakes = 0, nmouth = 0;
    {
        int last_dense = 0, last_flag = 0;
        DMG_HIP(hipMemcpyAsync(&last_dense, dense.i() + (N - 1), 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemcpyAsync(&last_flag, sflag.i() + (N - 1), 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemcpyAsync(&nmouth, nsel.p, 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        nlakes = last_dense + last_flag;
    }
    if (ctx->flow_hint_cells == N || nmouth == 0) { ctx->flow_hint_no_mouths = nmouth == 0; if (nmouth == 0) ctx->flow_hint_cells = N; }
    std::vector<int> sink_cell;                              // on the host only if a growth will read it (below, once the mouths are counted)
    if (nlakes) {
        DMG_ALLOC_VAR(sinkc, (size_t)nlakes * 4);
        hipLaunchKernelGGL(k_flow_sinkcells, dim3(nb), dim3(256), 0, st, label.i(), dense.i(), sinkc.i(), N);
    }

    // ---- the upstream sums, phase A (k_flow_local_keys has the story): every cell's sum inside its own catchment needs the labels, the
    // distances to the sinks and the receiver bits -- all there once the lakes are counted -- and nothing the pass search or the host's
    // growth produce.  It runs on two streams of its own (the large catchments on one, the small kinds on the other) beside the pass
    // candidates' sorts -- small launches that leave most of the chip idle -- and beside the host's growth; the main stream takes its sums
    // over (phase_a.join) before the first kernel that reads them.
    static const bool by_levels = dmg_hook("DMG_FLOW_LEVELS") != nullptr;      // experiments: the level-synchronous form
    static const int two_env = [] { const char *e = dmg_hook("DMG_FLOW_TWO_PHASE"); return e ? atoi(e) : 1; }();   // 0: one pass over all cells in order of their global depth (through round 5)
    // (phase A's keys hold a lake's dense id above FLOW_LOCAL_BITS of distance, in 32 bits, all ones being "no catchment": four million lakes.
    // The terrain the tectonics pass leaves -- every cell a lake of its own, 134 M of them at 16384x8192 -- takes the one pass.)
    bool two_phase = two_env != 0 && !by_levels && nmouth > 0 && (long long)nlakes < (1ll << (32 - FLOW_LOCAL_BITS)) - 1;      // (no mouth: nothing is reached, nothing to sum)
    dev_buf acc, keyA, keyA2, idxA, orderA, sortA, tickA, lakeoff, smaskA, nextB, bigA, midA;
    DMG_ALLOC(acc, N * 4);
    DMG_ALLOC(tickA, 32);                                // [0] ticket A, [1] error A, [2] cells with a catchment, [3] their largest distance to their sink, [4] ticket B, [5] error B, [6] cells of phase B
    DMG_HIP(hipMemsetAsync(tickA.p, 0, 32, st));
    // whatever way the solve is left, the side streams are done with the solve's buffers before those go back to the pool
    struct side_streams {
        dmg_ctx *ctx; hipStream_t main; bool forked = false, joined = false;
        int join()
        {
            if (!forked || joined) return DMG_OK;
            joined = true;
            DMG_HIP(hipStreamWaitEvent(main, ctx->flow_join, 0));
            DMG_HIP(hipStreamWaitEvent(main, ctx->flow_join2, 0));
            return DMG_OK;
        }
        ~side_streams() { if (forked && !joined) { (void)hipStreamSynchronize(ctx->flow_side); (void)hipStreamSynchronize(ctx->flow_side2); } }
    } phase_a{ ctx, st };
    if (two_phase) {
        DMG_ALLOC(keyA, N * 4); DMG_ALLOC(keyA2, N * 4); DMG_ALLOC(idxA, N * 4); DMG_ALLOC(orderA, N * 4);
        if (!ctx->flow_side) {
            DMG_HIP(hipStreamCreateWithFlags(&ctx->flow_side, hipStreamNonBlocking));
            DMG_HIP(hipStreamCreateWithFlags(&ctx->flow_side2, hipStreamNonBlocking));
            DMG_HIP(hipEventCreateWithFlags(&ctx->flow_fork, hipEventDisableTiming));
            DMG_HIP(hipEventCreateWithFlags(&ctx->flow_fork2, hipEventDisableTiming));
            DMG_HIP(hipEventCreateWithFlags(&ctx->flow_join, hipEventDisableTiming));
            DMG_HIP(hipEventCreateWithFlags(&ctx->flow_join2, hipEventDisableTiming));
        }
        hipStream_t sa = ctx->flow_side, sb2 = ctx->flow_side2;
        DMG_HIP(hipEventRecord(ctx->flow_fork, st));
        DMG_HIP(hipStreamWaitEvent(sa, ctx->flow_fork, 0));
        phase_a.forked = true;
        DMG_HIP(hipMemsetAsync(acc.p, 0xFF, N * 4, sa));                  // FLOW_NOT_READY everywhere
        if (nlakes > 0) {
            const unsigned nbk = (unsigned)((N + (size_t)256 * FLOW_ITEMS - 1) / ((size_t)256 * FLOW_ITEMS));
            hipLaunchKernelGGL(k_flow_local_keys, dim3(nbk), dim3(256), 0, sa, label.i(), dense.i(), dst[cur].i(), (unsigned *)keyA.p, (unsigned *)idxA.p,
                               tickA.i() + 2, tickA.i() + 3, N);
            size_t sb = 0;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sb, (const unsigned *)keyA.p, (unsigned *)keyA2.p, (const unsigned *)idxA.p, (unsigned *)orderA.p, (int)N, 0, 32, sa);
            DMG_ALLOC(sortA, sb + 256);
            // (cells without a catchment carry all ones: with the key's upper bits left out of the sort they would mix with the last lakes' --
            // so the sort takes all 32 bits where the lakes need 22)
            DMG_HIP(hipcub::DeviceRadixSort::SortPairs(sortA.p, sb, (const unsigned *)keyA.p, (unsigned *)keyA2.p, (const unsigned *)idxA.p, (unsigned *)orderA.p, (int)N, 0, 32, sa));
            DMG_ALLOC(lakeoff, (size_t)(nlakes + 1) * 4);
            DMG_ALLOC_VAR(bigA, (size_t)(nlakes + 1) * 4); DMG_ALLOC_VAR(midA, (size_t)(nlakes + 1) * 4);
            DMG_HIP(hipMemsetAsync(bigA.p, 0, 4, sa));
            DMG_HIP(hipMemsetAsync(midA.p, 0, 4, sa));
            DMG_ALLOC(smaskA, N * 2);
            hipLaunchKernelGGL(k_flow_lake_offsets, dim3(nb), dim3(256), 0, sa, (const unsigned *)keyA2.p, orderA.i(), (const unsigned short *)mask.p,
                               (const int *)(tickA.i() + 2), lakeoff.i(), (unsigned short *)smaskA.p, nlakes, N);
            hipLaunchKernelGGL(k_flow_lake_kinds, dim3((unsigned)((nlakes + 255) / 256)), dim3(256), 0, sa, lakeoff.i(), nlakes, bigA.i(), midA.i());
            // the large catchments (a few hundred workgroups, bound by the steps of the deepest) and the small ones (tens of thousands of
            // waves and lanes) do not wait for each other
            DMG_HIP(hipEventRecord(ctx->flow_fork2, sa));
            DMG_HIP(hipStreamWaitEvent(sb2, ctx->flow_fork2, 0));
            hipLaunchKernelGGL(k_flow_catchment_sums<1>, dim3((unsigned)std::min(nlakes, 2 * ctx->num_cu)), dim3(1024), 0, sa, fc, (const unsigned *)keyA2.p, orderA.i(),
                               lakeoff.i(), (const unsigned short *)smaskA.p, down.f(), (unsigned *)acc.p, (const int *)bigA.i());
            hipLaunchKernelGGL(k_flow_catchment_sums<0>, dim3((unsigned)std::min((nlakes + 3) / 4, 32 * ctx->num_cu)), dim3(256), 0, sb2, fc,
                               (const unsigned *)keyA2.p, orderA.i(), lakeoff.i(), (const unsigned short *)smaskA.p, down.f(), (unsigned *)acc.p, (const int *)midA.i());
            hipLaunchKernelGGL(k_flow_tiny_catchment_sums, dim3((unsigned)((nlakes + 255) / 256)), dim3(256), 0, sb2, fc, orderA.i(), lakeoff.i(),
                               (const unsigned short *)smaskA.p, down.f(), (unsigned *)acc.p, nlakes);
        } else {
            DMG_HIP(hipEventRecord(ctx->flow_fork2, sa));
            DMG_HIP(hipStreamWaitEvent(sb2, ctx->flow_fork2, 0));
        }
        DMG_HIP(hipEventRecord(ctx->flow_join, sa));
        DMG_HIP(hipEventRecord(ctx->flow_join2, sb2));
        DMG_HIP(hipGetLastError());
    }

    // boundary cells -> pass candidates; reduce to the lowest pass per lake pair on the device:
    // sort by (lake, from), keep each run's lowest (height, s), then order each lake's passes by height
    int cap = (int)std::min<size_t>(N, (size_t)1 << 28);
    dev_buf k0, v0, k1, v1, tmpst, sdist;
    int ncand = 0;
    if (nmouth) {
        DMG_ALLOC_VAR(k0, (size_t)cap * 8); DMG_ALLOC_VAR(v0, (size_t)cap * 8); DMG_ALLOC_VAR(k1, (size_t)cap * 8); DMG_ALLOC_VAR(v1, (size_t)cap * 8);
        DMG_HIP(hipMemsetAsync(count.p, 0, 4, st));
        hipLaunchKernelGGL(k_flow_passes, dim3((unsigned)((W + 255) / 256), (unsigned)((H + FLOW_ITEMS - 1) / FLOW_ITEMS)), dim3(256), 0, st, fc, d_h,    // tiles of 256 x FLOW_ITEMS cells
                           label.i(), dense.i(), (const unsigned short *)mask.p,
                           (unsigned long long *)k0.p, (unsigned long long *)v0.p, count.i(), cap);
        DMG_HIP(hipGetLastError());
        DMG_HIP(hipMemcpyAsync(&ncand, count.p, 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemsetAsync(nsel.p, 0, 8, st));           // (the select below counts into it)
        DMG_HIP(hipStreamSynchronize(st));
        if (ncand > cap) return dmg_fail(DMG_ENOMEM, "flow: %d pass candidates exceed the buffer", ncand);
    }
    std::vector<int> mouths(nmouth);
    dev_buf mouths_s, mouths_tmp;
    if (nmouth) {
        // the list comes out of k_flow_mouths in the order the blocks took their slots; the growth wants it ascending.  Sorted on
        // the device (std::sort of 30-46 k ids was 0.7 ms of a 7.6 ms config-2 iteration, more than the two pass sorts together)
        int bits = 1;
        while ((1ll << bits) < (long long)nlakes + 1) bits++;
        size_t mb = 0;
        DMG_ALLOC_VAR(mouths_s, (size_t)nmouth * 4);
        (void)hipcub::DeviceRadixSort::SortKeys(nullptr, mb, (const unsigned *)mouths_d.p, (unsigned *)mouths_s.p, nmouth, 0, bits, st);
        DMG_ALLOC_VAR(mouths_tmp, mb + 256);
        DMG_HIP(hipcub::DeviceRadixSort::SortKeys(mouths_tmp.p, mb, (const unsigned *)mouths_d.p, (unsigned *)mouths_s.p, nmouth, 0, bits, st));
        DMG_HIP(hipMemcpyAsync(mouths.data(), mouths_s.p, (size_t)nmouth * 4, hipMemcpyDeviceToHost, st));
    }
    if (nmouth && ncand && nlakes) {                         // the growth's `Nthbit(c.from,10)` test (:552) reads the lakes' sink cells
        sink_cell.resize((size_t)nlakes);
        DMG_HIP(hipMemcpyAsync(sink_cell.data(), sinkc.p, (size_t)nlakes * 4, hipMemcpyDeviceToHost, st));
    }

    int nuniq = 0;
    struct Entry { float h; int from, s, lake; };                     // = the int4 k_flow_entries writes
    struct Span { int cur, end; };                                     // per lake: next unread entry, end of its entries
    // the growth below visits these two tables in random order, a few hundred megabytes of them: transparent huge pages
    // (where the host allows them) take the TLB miss out of every visit
    std::vector<Entry, thp_allocator<Entry>> ent;
    std::vector<Span, thp_allocator<Span>> span((ncand && nmouth) ? (size_t)nlakes + 1 : (size_t)0, Span{ 0, 0 });   // (no mouths: no growth, no tables)
    dev_buf dent, dspan;
    // Without a river mouth the growth has nowhere to start (solvingConnections, :533-595, begins at the mouths): no pass is
    // ever looked at, so the reduction of the candidates to lake pairs is skipped.  Config 5 meets this case every
    // iteration: the terrain the tectonics render pass leaves is the plate map (a handful of flat values, Tectonics.cpp:590),
    // on which every cell is a lake of its own -- 134 M candidates at 16384x8192, 0.37 s of sorting that led nowhere.
    if (ncand && nmouth) {
        typedef unsigned long long u64;
        // (keys are lake << 32 | from, both dense ids: the sorts stop at the bits they have)
        int pair_bits = 33;
        while (pair_bits < 64 && (1ull << (pair_bits - 32)) < (unsigned long long)nlakes + 1) pair_bits++;
        size_t tb = 0, tb2 = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (u64 *)k0.p, (u64 *)k1.p, (u64 *)v0.p, (u64 *)v1.p, ncand, 0, pair_bits, st);
        (void)hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, (u64 *)k1.p, (u64 *)k0.p, (u64 *)v1.p, (u64 *)v0.p, nsel.i(), hipcub::Min(), ncand, st);
        DMG_ALLOC_VAR(tmpst, std::max(tb, tb2) + 256);
        // 1) by (lake, from): (k0,v0) -> (k1,v1)
        DMG_HIP(hipcub::DeviceRadixSort::SortPairs(tmpst.p, tb, (u64 *)k0.p, (u64 *)k1.p, (u64 *)v0.p, (u64 *)v1.p, ncand, 0, pair_bits, st));
        // 2) per lake pair its lowest (height, s) -- the smallest value of the run: (k1,v1) -> (k0,v0), one entry per pair.  (Through round 6
        // this was a sort by (height, s), a stable sort by (lake, from) over all 64 bits, segment heads and two selects: sixteen radix
        // passes over the candidates where seven and a reduction do.)
        DMG_HIP(hipcub::DeviceReduce::ReduceByKey(tmpst.p, tb2, (u64 *)k1.p, (u64 *)k0.p, (u64 *)v1.p, (u64 *)v0.p, nsel.i(), hipcub::Min(), ncand, st));
        DMG_HIP(hipMemcpyAsync(&nuniq, nsel.p, 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        // 3) (lake, height) order with (from, s) payload; stable, so equal heights stay in `from` order: packed in place, (k0,v0) -> (k1,v1)
        hipLaunchKernelGGL(k_flow_pack, dim3((nuniq + 255) / 256), dim3(256), 0, st, (const u64 *)k0.p, (const u64 *)v0.p,
                           (u64 *)k0.p, (u64 *)v0.p, nuniq);
        DMG_HIP(hipcub::DeviceRadixSort::SortPairs(tmpst.p, tb, (u64 *)k0.p, (u64 *)k1.p, (u64 *)v0.p, (u64 *)v1.p, nuniq, 0, pair_bits, st));
        DMG_ALLOC_VAR(sdist, (size_t)nuniq * 4);
        hipLaunchKernelGGL(k_flow_gather_dist, dim3((nuniq + 255) / 256), dim3(256), 0, st, (const u64 *)v1.p, dst[cur].i(), sdist.i(), nuniq);
        // the host's view: entries as 16-byte records and every lake's range of them (the sink distances stay on the device)
        DMG_ALLOC_VAR(dent, (size_t)std::max(nuniq, 1) * 16); DMG_ALLOC_VAR(dspan, (size_t)(nlakes + 1) * 8);
        DMG_HIP(hipMemsetAsync(dspan.p, 0, (size_t)(nlakes + 1) * 8, st));
        hipLaunchKernelGGL(k_flow_entries, dim3((nuniq + 255) / 256), dim3(256), 0, st, (const u64 *)k1.p, (const u64 *)v1.p, (int4 *)dent.p, (int2 *)dspan.p, nuniq);
        DMG_HIP(hipGetLastError());
        ent.resize(nuniq);
        DMG_HIP(hipMemcpyAsync(ent.data(), dent.p, (size_t)nuniq * 16, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemcpyAsync(span.data(), dspan.p, (size_t)(nlakes + 1) * 8, hipMemcpyDeviceToHost, st));
    }
    // (phase A of the upstream sums has been running on the side streams since the lakes were counted.)  The host waits for the entries
    // only -- an event behind their copies -- and what the stages after the growth need cleared or preset is queued behind that event:
    // done while the host grows the lake graph, not after it
    if (!ctx->flow_copied) DMG_HIP(hipEventCreateWithFlags(&ctx->flow_copied, hipEventDisableTiming));
    DMG_HIP(hipEventRecord(ctx->flow_copied, st));
    dev_buf dlo, dcf, gdepth, dstart, order, dmax, dlevel, dassigned, dpar0, dof0, dlv0, dmo;
    DMG_ALLOC(dlo, N * 4); DMG_ALLOC(dcf, N * 4); DMG_ALLOC(gdepth, N * 4); DMG_ALLOC(dmax, 8);
    DMG_HIP(hipMemsetAsync(dlo.p, 0xFF, N * 4, st));         // -1
    DMG_HIP(hipMemsetAsync(dcf.p, 0xFF, N * 4, st));
    DMG_HIP(hipMemsetAsync(dmax.p, 0, 8, st));
    if (nlakes) {
        const unsigned lb0 = (unsigned)((nlakes + 255) / 256);
        DMG_ALLOC_VAR(dpar0, (size_t)nlakes * 4); DMG_ALLOC_VAR(dof0, (size_t)nlakes * 4); DMG_ALLOC_VAR(dlv0, (size_t)nlakes * 4); DMG_ALLOC_VAR(dmo, (size_t)nlakes);
        hipLaunchKernelGGL(k_flow_lake_init, dim3(lb0), dim3(256), 0, st, dpar0.i(), nlakes);
        DMG_HIP(hipMemsetAsync(dof0.p, 0, (size_t)nlakes * 4, st));
        DMG_HIP(hipMemsetAsync(dlv0.p, 0, (size_t)nlakes * 4, st));
        DMG_HIP(hipMemsetAsync(dmo.p, 0, (size_t)nlakes, st));
        if (nmouth) hipLaunchKernelGGL(k_flow_mark, dim3((nmouth + 255) / 256), dim3(256), 0, st, mouths_d.i(), (unsigned char *)dmo.p, nmouth);
    }
    DMG_HIP(hipMemsetAsync(tmp.p, 0xFF, N * 4, st));         // (k_flow_lastpos: last position per cell, -1)
    dev_buf marks;
    if (two_phase) {
        DMG_ALLOC(marks, N * 4); DMG_ALLOC(nextB, N * 4);
        DMG_HIP(hipMemsetAsync(marks.p, 0, N * 4, st));
        DMG_HIP(hipMemsetAsync(nextB.p, 0xFF, N * 4, st));
    }
    hipLaunchKernelGGL(k_fill_f, dim3(nb), dim3(256), 0, st, d_flow, -1.0f, N);        // unreached cells keep -1 (:598-604)
    DMG_HIP(hipGetLastError());
    DMG_HIP(hipEventSynchronize(ctx->flow_copied));
    lap("passes: sort/reduce/download");
    if (timing) fprintf(stderr, "[flow] %d lakes, %d pass candidates -> %d lake pairs, %d river mouths\n", nlakes, ncand, nuniq, nmouth);

    // solvingConnections (:533-595) on the reduced list, all lakes by dense id: entry i = pass into lake `lake` at
    // cell s from lake `from`; entries are sorted by (lake, height, from) -- built on the device (k_flow_entries)
    lap("host: entry table");
    std::vector<int> assigned;                                         // connection events in order: connections[s] = entry
    const bool mouth_filter_on = !(lakes & DMG_FLOW_FIX_MOUTH_FILTER);
    // which growth: one host thread (small graphs: nothing to gain), or the cascades on all cores (k_growth_* above)
    static const int growth_env = [] { const char *e = dmg_hook("DMG_FLOW_GROWTH"); return !e ? 0 : (!strcmp(e, "serial") ? 1 : (!strcmp(e, "parallel") ? 2 : 0)); }();
    const bool parallel_growth = nuniq > 0 && nmouth > 0 && (growth_env == 2 || (growth_env == 0 && nlakes >= 100000));
    if (parallel_growth) {
        typedef unsigned long long u64;
        // (1) rank of every pass in the growth's total order (h, from, s): two stable radix sorts
        dev_buf gks, gks2, gidx, gidx2, gkhf, gkhf2, gorder, grank, gd, gq, gfa, gfb, gcnt, gvs, gtmp, gds, gls, glid, ghead;
        DMG_ALLOC_VAR(gks, (size_t)nuniq * 4); DMG_ALLOC_VAR(gks2, (size_t)nuniq * 4); DMG_ALLOC_VAR(gidx, (size_t)nuniq * 4); DMG_ALLOC_VAR(gidx2, (size_t)nuniq * 4);
        DMG_ALLOC_VAR(gkhf, (size_t)nuniq * 8); DMG_ALLOC_VAR(gkhf2, (size_t)nuniq * 8); DMG_ALLOC_VAR(gorder, (size_t)nuniq * 4); DMG_ALLOC_VAR(grank, (size_t)nuniq * 4);
        const unsigned eb = (unsigned)((nuniq + 255) / 256), lb = (unsigned)((nlakes + 255) / 256);
        hipLaunchKernelGGL(k_growth_keys, dim3(eb), dim3(256), 0, st, (const int4 *)dent.p, (unsigned *)gks.p, gidx.i(), nuniq);
        size_t t1 = 0, t2 = 0, t3 = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (unsigned *)gks.p, (unsigned *)gks2.p, gidx.i(), gidx2.i(), nuniq, 0, 32, st);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, (u64 *)gkhf.p, (u64 *)gkhf2.p, gidx2.i(), gorder.i(), nuniq, 0, 64, st);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t3, (unsigned *)gks.p, (unsigned *)gks2.p, gidx.i(), gidx2.i(), nlakes, 0, 32, st);
        DMG_ALLOC_VAR(gtmp, std::max(t1, std::max(t2, t3)) + 256);
        DMG_HIP(hipcub::DeviceRadixSort::SortPairs(gtmp.p, t1, (unsigned *)gks.p, (unsigned *)gks2.p, gidx.i(), gidx2.i(), nuniq, 0, 32, st));      // by s
        hipLaunchKernelGGL(k_growth_keys2, dim3(eb), dim3(256), 0, st, (const int4 *)dent.p, gidx2.i(), (u64 *)gkhf.p, nuniq);
        DMG_HIP(hipcub::DeviceRadixSort::SortPairs(gtmp.p, t2, (u64 *)gkhf.p, (u64 *)gkhf2.p, gidx2.i(), gorder.i(), nuniq, 0, 64, st));           // stable by (h, from)
        hipLaunchKernelGGL(k_growth_rank, dim3(eb), dim3(256), 0, st, gorder.i(), (unsigned *)grank.p, nuniq);
        DMG_HIP(hipGetLastError());
        // (2) where a lake's usable passes start: a river mouth's first look at its list drops the passes in front of the
        // first one it can use -- neighbours that are mouths themselves and, with the reference's `Nthbit(c.from,10)` (:552),
        // neighbours whose sink cell has bit 512 set -- for good (push_next(lake, initial) above)
        std::vector<char> placed(nlakes, 0);
        for (int m : mouths) placed[m] = 1;
        std::vector<int> vstart((size_t)nlakes);
        for (int l = 0; l < nlakes; l++) vstart[l] = span[l].cur;
        for (int m : mouths) {
            int i = span[m].cur;
            for (; i < span[m].end; i++) {
                if (placed[ent[i].from]) continue;
                if (mouth_filter_on && (sink_cell[ent[i].from] & 512)) continue;
                break;
            }
            vstart[m] = i;
        }
        if ((rc = upload_vec(ctx, gvs, vstart))) return rc;
        // (3) minimax levels: frontier relaxation from the mouths until nothing falls any more
        DMG_ALLOC_VAR(gd, (size_t)nlakes * 4); DMG_ALLOC_VAR(gq, (size_t)nlakes * 4); DMG_ALLOC_VAR(gfa, (size_t)nlakes * 4); DMG_ALLOC_VAR(gfb, (size_t)nlakes * 4); DMG_ALLOC(gcnt, 12);
        hipLaunchKernelGGL(k_growth_init, dim3(lb), dim3(256), 0, st, (unsigned *)gd.p, gq.i(), nlakes);
        hipLaunchKernelGGL(k_growth_seed, dim3((unsigned)((nmouth + 255) / 256)), dim3(256), 0, st, mouths_d.i(), nmouth, (unsigned *)gd.p, gfa.i(), gcnt.i());
        int rounds = 0;
        for (bool more = true; more;) {
            for (int k = 0; k < 16; k++, rounds++) {
                const int in = rounds & 1;                                     // (the frontier lists alternate, the counters go round in threes)
                hipLaunchKernelGGL(k_growth_relax, dim3(1024), dim3(256), 0, st, in ? gfb.i() : gfa.i(), gcnt.i() + rounds % 3, in ? gfa.i() : gfb.i(), gcnt.i() + (rounds + 1) % 3,
                                   (unsigned *)gd.p, (const int2 *)dspan.p, gvs.i(), (const int4 *)dent.p, (const unsigned *)grank.p, gq.i(), rounds, gcnt.i() + (rounds + 2) % 3);
            }
            int cnt[3] = { 0, 0, 0 };
            DMG_HIP(hipMemcpyAsync(cnt, gcnt.p, 12, hipMemcpyDeviceToHost, st));
            DMG_HIP(hipStreamSynchronize(st));
            more = cnt[rounds % 3] != 0;
            if (rounds > 4 * nlakes + 64) return dmg_fail(DMG_ESTATE, "flow: the minimax levels of the lake graph did not settle");
        }
        // (4) lakes in order of their level; the pass that opens each lake's cascade
        DMG_ALLOC_VAR(gds, (size_t)nlakes * 4); DMG_ALLOC_VAR(glid, (size_t)nlakes * 4); DMG_ALLOC_VAR(gls, (size_t)nlakes * 4); DMG_ALLOC_VAR(ghead, (size_t)nlakes * 4);
        hipLaunchKernelGGL(k_growth_iota, dim3(lb), dim3(256), 0, st, glid.i(), nlakes);
        DMG_HIP(hipcub::DeviceRadixSort::SortPairs(gtmp.p, t3, (unsigned *)gd.p, (unsigned *)gds.p, glid.i(), gls.i(), nlakes, 0, 32, st));
        hipLaunchKernelGGL(k_growth_heads, dim3(lb), dim3(256), 0, st, (const unsigned *)gds.p, gorder.i(), ghead.i(), nlakes, nuniq);
        DMG_HIP(hipGetLastError());
        std::vector<unsigned> dl((size_t)nlakes), dsorted((size_t)nlakes);
        std::vector<int> lsorted((size_t)nlakes), head((size_t)nlakes);
        DMG_HIP(hipMemcpyAsync(dl.data(), gd.p, (size_t)nlakes * 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemcpyAsync(dsorted.data(), gds.p, (size_t)nlakes * 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemcpyAsync(lsorted.data(), gls.p, (size_t)nlakes * 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipMemcpyAsync(head.data(), ghead.p, (size_t)nlakes * 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        lap("growth: ranks + levels (device)");
        // (5) the cascades, on all cores.  Sorted positions [p0, p1): the mouths (level 0), then the cascades by level, then
        // the lakes no pass reaches.  A cascade of n lakes makes n connections, so its place in `assigned` is its place here.
        size_t p0 = 0, p1 = (size_t)nlakes;
        while (p0 < p1 && dsorted[p0] == 0u) p0++;
        while (p1 > p0 && dsorted[p1 - 1] == 0xFFFFFFFFu) p1--;
        assigned.assign(p1 - p0, -1);
        // every thread takes chunks of sorted positions from a counter and handles the cascades that START in its chunk (a
        // cascade may reach into the next chunk: its lakes are contiguous whatever the chunking)
        const size_t CHUNK = 2048;
        std::atomic<size_t> next{p0};
        std::atomic<int> bad{0};
        std::atomic<long long> nbig{0};
        struct Cand { float h; int from, s, idx, lake; };
        auto worse = [](const Cand &a, const Cand &b) { return std::tie(a.h, a.from, a.s) > std::tie(b.h, b.from, b.s); };
        auto work = [&]() {
            std::vector<Cand> heap;
            long long mybig = 0;
            for (;;) {
                const size_t c0 = next.fetch_add(CHUNK);
                if (c0 >= p1) break;
                const size_t c1 = std::min(c0 + CHUNK, p1);
                for (size_t q = c0; q < c1;) {
                    const unsigned x = dsorted[q];
                    if (q > p0 && dsorted[q - 1] == x) { q++; continue; }       // inside a cascade that started earlier
                    size_t e = q + 1;
                    while (e < p1 && dsorted[e] == x) e++;
                    const int ex = head[q];
                    if (e - q == 1) { assigned[q - p0] = ex; if (ex < 0) bad++; q = e; continue; }
                    mybig++;
                    if (ex < 0 || dl[ent[ex].from] != x) { bad++; q = e; continue; }
                    heap.clear();
                    size_t pos = q - p0;
                    auto push_next = [&](int lake) {
                        Span &sp = span[lake];
                        while (sp.cur < sp.end) {
                            const int i = sp.cur++;
                            const int f = ent[i].from;
                            if (dl[f] != x || placed[f]) continue;         // (another cascade's lake is never looked at: its flag is another thread's)
                            heap.push_back(Cand{ ent[i].h, f, ent[i].s, i, lake });
                            std::push_heap(heap.begin(), heap.end(), worse);
                            break;
                        }
                    };
                    const int f0 = ent[ex].from;
                    placed[f0] = 1;
                    assigned[pos++] = ex;
                    push_next(f0);
                    while (!heap.empty()) {
                        std::pop_heap(heap.begin(), heap.end(), worse);
                        const Cand top = heap.back();
                        heap.pop_back();
                        if (placed[top.from]) { push_next(top.lake); continue; }
                        placed[top.from] = 1;
                        if (pos >= e - p0) { bad++; break; }
                        assigned[pos++] = top.idx;
                        push_next(top.lake);
                        push_next(top.from);
                    }
                    if (pos != e - p0) bad++;
                    q = e < c1 ? e : c1;
                }
            }
            nbig += mybig;
        };
        const unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < nthreads; k++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (bad.load()) return dmg_fail(DMG_ESTATE, "flow: the parallel growth of the lake graph lost a lake (%d cascades inconsistent)", bad.load());
        if (timing) fprintf(stderr, "[flow] parallel growth: %d relaxation rounds, %zu connections, %zu cascades with more than one lake, %u threads\n", rounds, assigned.size(),
                            (size_t)nbig.load(), nthreads);
    } else if (nuniq > 0 && nmouth > 0) {
    // the heap carries its keys (16 B) instead of indices into `ent`: a sift touches only the heap's own cache lines
    struct Cand { float h; int from, s, idx, lake; };                  // (lake: of the entry, so that a pop need not look it up)
    auto worse = [](const Cand &a, const Cand &b) {                    // priority_queue keeps the LOWEST (h, from, s) on top
        return std::tie(a.h, a.from, a.s) > std::tie(b.h, b.from, b.s);
    };
    // The candidates, lowest (h, from, s) first.  One binary heap over a million 16-byte entries costs a pop ~8 cache
    // misses on its way down (the growth spent most of its time there); here the candidates are spread over 2^18 buckets by
    // the top bits of their height (a monotone key: equal heights share a bucket), each bucket a small heap of its own, and a
    // three-level bitmap finds the lowest bucket in use: a pop stays inside one small, hot bucket.  Same total order, same
    // sequence of pops as std::set<pass, comp> in solvingConnections.
    struct cand_queue {
        const int BITS;                                                // 18, or 0 = one heap (small graphs: a vector per bucket costs more than it saves)
        std::vector<std::vector<Cand>> bucket;
        std::vector<unsigned long long> l0, l1;
        unsigned long long l2 = 0;
        size_t count = 0;
        decltype(worse) cmp;
        cand_queue(decltype(worse) c, int bits) : BITS(bits), bucket((size_t)1 << bits), l0(bits ? (size_t)1 << (bits - 6) : 1, 0), l1(bits ? (size_t)1 << (bits - 12) : 1, 0), cmp(c) {}
        unsigned key(float h) const
        {
            if (!BITS) return 0;
            if (h == 0.0f) h = 0.0f;                                   // -0.0 compares equal to +0.0: one bucket
            unsigned u;
            memcpy(&u, &h, 4);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // monotone in h
            return u >> (32 - BITS);
        }
        bool empty() const { return count == 0; }
        void push(const Cand &c)
        {
            const unsigned b = key(c.h);
            std::vector<Cand> &v = bucket[b];
            v.push_back(c);
            std::push_heap(v.begin(), v.end(), cmp);
            l0[b >> 6] |= 1ull << (b & 63); l1[b >> 12] |= 1ull << ((b >> 6) & 63); l2 |= 1ull << (b >> 12);
            count++;
        }
        unsigned lowest() const
        {
            if (!BITS) return 0;
            const unsigned i2 = (unsigned)__builtin_ctzll(l2);
            const unsigned i1 = (i2 << 6) | (unsigned)__builtin_ctzll(l1[i2]);
            return (i1 << 6) | (unsigned)__builtin_ctzll(l0[i1]);
        }
        const Cand &top() const { return bucket[lowest()].front(); }
        void pop()
        {
            const unsigned b = lowest();
            std::vector<Cand> &v = bucket[b];
            std::pop_heap(v.begin(), v.end(), cmp);
            v.pop_back();
            count--;
            if (v.empty()) {
                if (!(l0[b >> 6] &= ~(1ull << (b & 63))) && !(l1[b >> 12] &= ~(1ull << ((b >> 6) & 63)))) l2 &= ~(1ull << (b >> 12));
            }
        }
    } candidates(worse, nlakes >= (1 << 20) ? 18 : 0);           // (a vector per bucket: 2^18 of them only where millions of lakes fill them)
    std::vector<char> placed(nlakes, 0);
    for (int m : mouths) placed[m] = 1;
    // The growth hops all over the map, so each visit of a lake is a chain of cache misses (its span, its first unread
    // entry, the flag of that entry's lake).  The lake a candidate would place is known when the candidate is pushed:
    // its span is prefetched then, and its first entry when the candidate is popped, one lake's worth of work ahead.
    const bool mouth_filter = mouth_filter_on;
    auto push_next = [&](int lake, bool initial = false) {
        Span &sp = span[lake];
        while (sp.cur < sp.end) {
            int i = sp.cur++;
            if (placed[ent[i].from]) continue;
            // `if (Nthbit(c.from,10)) continue;` (:552): bit 10 of the neighbouring lake's cell index, first stage only
            if (initial && mouth_filter && (sink_cell[ent[i].from] & 512)) continue;
            candidates.push(Cand{ ent[i].h, ent[i].from, ent[i].s, i, lake });
            __builtin_prefetch(&span[ent[i].from]);
            break;
        }
    };
    const bool far_tables = (size_t)nuniq * sizeof(Entry) > ((size_t)32 << 20);
    std::vector<int> stat_depth;
    int stat_maxdepth = 0, stat_from = 0, stat_s = 0; long long stat_ncascade = 0, stat_cur = 0, stat_maxcascade = 0, stat_multi = 0, stat_inmulti = 0;
    float stat_h = 0.0f; bool stat_have = false;
    const auto tg0 = std::chrono::steady_clock::now();
    for (int lake : mouths) push_next(lake, true);
    const auto tg1 = std::chrono::steady_clock::now();
    long long npop = 0, npush0 = (long long)candidates.count;
    while (!candidates.empty()) {
        npop++;
        const Cand top = candidates.top();
        candidates.pop();
        const int c = top.idx;
        // What the next pops will need.  The growth is a chain of dependent cache misses (a lake's entry range, its next
        // unread entry, the flag of that entry's lake), ~15 M of them; the candidates at the top of the lowest bucket's
        // heap are (mostly) the next ones to be popped: their ranges are requested two to six pops ahead, and the entries
        // those ranges point at one to two pops ahead
        if (far_tables && !candidates.empty()) {                        // (tables that fit the caches need no help: the look-ahead costs ~30 ns a pop)
            const std::vector<Cand> &v = candidates.bucket[candidates.lowest()];
            const size_t nv = v.size();
            for (size_t k = 3; k < std::min<size_t>(nv, 7); k++) {
                __builtin_prefetch(&span[v[k].lake]); __builtin_prefetch(&span[v[k].from]); __builtin_prefetch(&placed[v[k].from]);
            }
            for (size_t k = 0; k < std::min<size_t>(nv, 3); k++) {
                __builtin_prefetch(&ent[span[v[k].lake].cur]); __builtin_prefetch(&ent[span[v[k].from].cur]);
            }
        }
        if (placed[top.from]) { push_next(top.lake); continue; }
        placed[top.from] = 1;
        if (timing) {                                                  // statistics for DESIGN.md (the growth's structure): hops and "cascades"
            if (stat_depth.empty()) stat_depth.assign((size_t)nlakes, 0);
            stat_depth[top.from] = stat_depth[top.lake] + 1;
            stat_maxdepth = std::max(stat_maxdepth, stat_depth[top.from]);
            const bool higher = !stat_have || std::tie(top.h, top.from, top.s) > std::tie(stat_h, stat_from, stat_s);
            if (higher) { stat_have = true; stat_h = top.h; stat_from = top.from; stat_s = top.s; stat_ncascade++; stat_cur = 0; }
            stat_cur++;
            stat_maxcascade = std::max(stat_maxcascade, stat_cur);
            if (stat_cur == 2) stat_multi++;
            if (stat_cur >= 2) stat_inmulti += stat_cur == 2 ? 2 : 1;
        }
        __builtin_prefetch(&ent[span[top.from].cur]);
        assigned.push_back(c);                                         // a later pass at the same cell replaces this one (:588)
        push_next(top.lake);                                           // (independent of the next line: disjoint lakes)
        push_next(top.from);
    }
    if (timing) fprintf(stderr, "[flow] growth on the host: the mouths' first passes %.3f ms (%lld pushed), the growth %.3f ms (%lld pops)\n", std::chrono::duration<double, std::milli>(tg1 - tg0).count(), npush0,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg1).count(), npop);
    if (timing) fprintf(stderr, "[flow] growth: %zu connections, deepest lake %d hops from its river mouth; %lld cascades (a new highest pass = a new cascade), "
                                "%lld of them with more than one lake holding %lld lakes, the largest %lld lakes\n", assigned.size(), stat_maxdepth, stat_ncascade, stat_multi, stat_inmulti, stat_maxcascade);
    }
    lap("host: prim growth");
    // connections[tolocation] = pass (:588): per tolocation cell the LAST assigned entry survives; its lake hangs one below
    // the cell it drains into (calculateflow's recursion: river mouths at depth 0).  Both on the device: the host only
    // hands over the order in which the growth made its connections
    if (nlakes) {
        const int na = (int)assigned.size();
        if (na) {
            typedef unsigned long long u64;
            if ((rc = upload_vec(ctx, dassigned, assigned))) return rc;
            hipLaunchKernelGGL(k_flow_lastpos, dim3((na + 255) / 256), dim3(256), 0, st, dassigned.i(), (const u64 *)v1.p, tmp.i(), na);
            hipLaunchKernelGGL(k_flow_connect, dim3((na + 255) / 256), dim3(256), 0, st, dassigned.i(), (const u64 *)k1.p, (const u64 *)v1.p, sdist.i(), tmp.i(),
                               sinkc.i(), dcf.i(), dpar0.i(), dof0.i(), dlv0.f(), na);
        }
        DMG_HIP(hipGetLastError());
    }
    lap("connections (device)");

    const bool draw_lakes = (lakes & DMG_FLOW_LAKES) != 0;
    if (draw_lakes) {
        DMG_ALLOC(dlevel, N * 4);
        DMG_HIP(hipMemsetAsync(dlevel.p, 0xFF, N * 4, st));  // NaN: compares false
    }
    if (nlakes) {
        dev_buf dp1, dof1, dlv1;
        DMG_ALLOC_VAR(dp1, (size_t)nlakes * 4); DMG_ALLOC_VAR(dof1, (size_t)nlakes * 4); DMG_ALLOC_VAR(dlv1, (size_t)nlakes * 4);
        int *dp[2] = { dpar0.i(), dp1.i() }, *dof[2] = { dof0.i(), dof1.i() };
        float *dlv[2] = { dlv0.f(), dlv1.f() };
        int rounds = 1;
        while ((1ll << rounds) < nlakes) rounds++;           // paths are at most nlakes long
        if (assigned.empty()) rounds = -1;                   // (no connection: every lake is its own root, a sweep is the identity -- 134 M lakes on the
                                                             // tectonics pass' plate map, five sweeps before the first question, 3 ms of config 5's iteration)
        int b = 0;
        const unsigned lb = (unsigned)((nlakes + 255) / 256);
        // The forest is far shallower than it is large (2.6 M lakes at 16384x8192 hang a few hundred deep): where a sweep costs more
        // than a read-back -- a quarter of a million lakes and more -- every other sweep from the fifth on reports whether a
        // pointer still moved, and the loop ends with the first that says none did.  (Same tables: further sweeps are the identity.)
        const bool ask = nlakes >= (1 << 18);
        for (int it = 0; it <= rounds; it++, b ^= 1) {
            const bool check = ask && it >= 4 && (it & 1) == 0 && it < 64;
            int *flag = changed.i() + 64 + it;           // (words 64.. of the table cleared in front of the cells' pointer doubling)
            hipLaunchKernelGGL(k_lake_jump, dim3(lb), dim3(256), 0, st, dp[b], dof[b], dlv[b], dp[b ^ 1], dof[b ^ 1], dlv[b ^ 1], nlakes,
                               check ? flag : (int *)nullptr);
            if (check) {
                int ch = 0;
                DMG_HIP(hipMemcpyAsync(&ch, flag, 4, hipMemcpyDeviceToHost, st));
                DMG_HIP(hipStreamSynchronize(st));
                if (!ch) { b ^= 1; break; }
            }
        }
        hipLaunchKernelGGL(k_lake_scatter, dim3(lb), dim3(256), 0, st, dp[b], dof[b], dlv[b], (const unsigned char *)dmo.p, sinkc.i(),
                           dlo.i(), draw_lakes ? dlevel.f() : nullptr, dmax.i() + 1, nlakes);
        DMG_HIP(hipGetLastError());
        // (no synchronisation here: the tables above go back to the context's pool, whose reuse is ordered by this stream, and the host
        // vectors they were filled from live until the solve returns)
    }
    if ((rc = phase_a.join())) return rc;                   // (phase A's sums, counts and lists: the main stream's from here on)
    hipLaunchKernelGGL(k_flow_gdepth, dim3(nbi), dim3(256), 0, st, label.i(), dst[cur].i(), dlo.i(), gdepth.i(), dmax.i(), N);
    DMG_HIP(hipGetLastError());
    // phase B's cells: marked from the connections' cells down to the sinks, listed with their global depths, their sums withdrawn
    if (two_phase) {
        hipLaunchKernelGGL(k_flow_mark_affected, dim3(nb), dim3(256), 0, st, fc, (const unsigned char *)code.p, dcf.i(), gdepth.i(), marks.i(), nextB.i(), N);
        hipLaunchKernelGGL(k_flow_collect_affected, dim3(nbi), dim3(256), 0, st, marks.i(), gdepth.i(), (unsigned *)acc.p, (unsigned *)keyA.p, (unsigned *)idxA.p, tickA.i() + 6, N);
        DMG_HIP(hipGetLastError());
    }
    int max_depth = 0, tick_host[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    DMG_HIP(hipMemcpyAsync(&max_depth, dmax.p, 4, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipMemcpyAsync(tick_host, tickA.p, 32, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    const int levels = max_depth + 1;
    int depth_bits = 1;
    while ((1 << depth_bits) <= levels) depth_bits++;
    if (two_phase && tick_host[1]) return dmg_fail(DMG_EHIP, "flow: the sums inside the catchments did not complete (a cell waited for a sum that was never published)");
    static const int local_limit = [] { const char *e = dmg_hook("DMG_FLOW_LOCAL_LIMIT"); return e ? atoi(e) : (1 << FLOW_LOCAL_BITS) - 1; }();      // tests: the fallback on small grids
    if (two_phase && tick_host[3] >= local_limit) {
        // (a catchment deeper than phase A's keys have bits for: the one-pass form, from scratch)
        two_phase = false;
        DMG_HIP(hipMemsetAsync(tickA.p, 0, 32, st));
    }
    static const int look_env = [] { const char *e = dmg_hook("DMG_FLOW_LOOK_THROUGH"); return e ? atoi(e) : -1; }();      // experiments: 0 = every cell waits for its sums to be published
    static const int nblocks_env = [] { const char *e = dmg_hook("DMG_FLOW_NBLOCKS"); return e ? atoi(e) : 0; }();
    if (two_phase) {
        const int n_aff = tick_host[6];
        if (timing) fprintf(stderr, "[flow] two phases: %d cells with a catchment (deepest %d from its sink), %d of them behind a connection; %d levels of global depth\n",
                            tick_host[2], tick_host[3], n_aff, levels);
        lap("upload + depth sort");
        static const int walk_env = [] { const char *e = dmg_hook("DMG_FLOW_WALK"); return e ? atoi(e) : 1; }();      // 0: phase B as a dataflow in depth order
        int cell_bits = 1;
        while ((1ull << cell_bits) < (unsigned long long)N) cell_bits++;
        if (n_aff > 0 && walk_env != 0 && 2 * depth_bits + cell_bits <= 64) {
            // the walk along the reaches (k_flow_walk): keys from the marks, the cells reach by reach, 64 to a wave
            dev_buf wkey, wkey2, sortB, top[2], hop[2], tail;
            DMG_ALLOC(wkey, (size_t)n_aff * 8); DMG_ALLOC(wkey2, (size_t)n_aff * 8);
            DMG_ALLOC(top[0], N * 4); DMG_ALLOC(top[1], N * 4); DMG_ALLOC(hop[0], N * 4); DMG_ALLOC(hop[1], N * 4); DMG_ALLOC(tail, N * 4);
            const dim3 lb((unsigned)((n_aff + 255) / 256));
            reach_tables rt;
            rt.cell = (const unsigned *)idxA.p; rt.gdepth = gdepth.i(); rt.next = nextB.i(); rt.marked = marks.i();
            rt.top[0] = top[0].i(); rt.top[1] = top[1].i(); rt.hop[0] = hop[0].i(); rt.hop[1] = hop[1].i(); rt.tail = tail.i();
            rt.key = (unsigned long long *)wkey.p;
            rt.n = n_aff; rt.rounds = depth_bits; rt.depth_bits = depth_bits; rt.cell_bits = cell_bits; rt.max_depth = max_depth;      // (2^depth_bits > the longest river)
            {
                hipLaunchKernelGGL(k_reach_init, lb, dim3(256), 0, st, rt);
                int from = 0;
                for (int r = 0; r < rt.rounds; r++, from ^= 1) hipLaunchKernelGGL(k_reach_height, lb, dim3(256), 0, st, rt, from);
                hipLaunchKernelGGL(k_reach_choose, lb, dim3(256), 0, st, rt, fc, (const unsigned short *)mask.p, dcf.i(), from);
                for (int r = 0; r < rt.rounds; r++) hipLaunchKernelGGL(k_reach_tail, lb, dim3(256), 0, st, rt);
                hipLaunchKernelGGL(k_flow_walk_keys, lb, dim3(256), 0, st, rt);
            }
            size_t sb = 0;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sb, (const unsigned long long *)wkey.p, (unsigned long long *)wkey2.p, (const unsigned *)idxA.p,
                                                     (unsigned *)orderA.p, n_aff, 0, 2 * depth_bits + cell_bits, st);
            DMG_ALLOC(sortB, sb + 256);
            DMG_HIP(hipcub::DeviceRadixSort::SortPairs(sortB.p, sb, (const unsigned long long *)wkey.p, (unsigned long long *)wkey2.p, (const unsigned *)idxA.p,
                                                       (unsigned *)orderA.p, n_aff, 0, 2 * depth_bits + cell_bits, st));
            const int waves = (n_aff + 63) / 64;
            const int blocks = std::min((waves + 3) / 4, nblocks_env > 0 ? nblocks_env : 256);
            hipLaunchKernelGGL(k_flow_walk, dim3(blocks), dim3(256), 0, st, fc, (const unsigned long long *)wkey2.p, orderA.i(), n_aff, depth_bits,
                               (const unsigned short *)mask.p, dcf.i(), down.f(), (unsigned *)acc.p, tickA.i() + 4, tickA.i() + 5);
            DMG_HIP(hipGetLastError());
            int flow_err = 0;
            DMG_HIP(hipMemcpyAsync(&flow_err, tickA.i() + 5, 4, hipMemcpyDeviceToHost, st));
            DMG_HIP(hipStreamSynchronize(st));
            if (flow_err) return dmg_fail(DMG_EHIP, flow_err & 2 ? "flow: the cells of a reach do not drain into each other (the marks and the sort disagree)"
                                                                 : "flow: the upstream sums did not complete (a cell waited for a sum that was never published)");
        } else if (n_aff > 0) {
            size_t sb = 0;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sb, (const unsigned *)keyA.p, (unsigned *)keyA2.p, (const unsigned *)idxA.p, (unsigned *)orderA.p, n_aff, 0, depth_bits, st);
            dev_buf sortB;
            DMG_ALLOC(sortB, sb + 256);
            DMG_HIP(hipcub::DeviceRadixSort::SortPairs(sortB.p, sb, (const unsigned *)keyA.p, (unsigned *)keyA2.p, (const unsigned *)idxA.p, (unsigned *)orderA.p, n_aff, 0, depth_bits, st));
            // (few cells, a long chain: the small grid and the look-through, as for the sparse levels of the one-pass form)
            const int waves = (n_aff + 63) / 64;
            const int blocks = std::min((waves + 3) / 4, nblocks_env > 0 ? nblocks_env : 256);
            hipLaunchKernelGGL(k_flow_dataflow, dim3(blocks), dim3(256), 0, st, fc, orderA.i(), n_aff, (const unsigned short *)mask.p, dcf.i(),
                               down.f(), (unsigned *)acc.p, tickA.i() + 4, tickA.i() + 5, look_env >= 0 ? look_env : 1, (const int *)nullptr);
            DMG_HIP(hipGetLastError());
            int flow_err = 0;
            DMG_HIP(hipMemcpyAsync(&flow_err, tickA.i() + 5, 4, hipMemcpyDeviceToHost, st));
            DMG_HIP(hipStreamSynchronize(st));
            if (flow_err) return dmg_fail(DMG_EHIP, "flow: the upstream sums did not complete (a cell waited for a sum that was never published)");
        }
    } else {
        DMG_ALLOC(order, N * 4);
        dev_buf sdepth, iota, sorttmp;
        DMG_ALLOC(sdepth, N * 4); DMG_ALLOC(iota, N * 4);
        hipLaunchKernelGGL(k_flow_iota, dim3(nb), dim3(256), 0, st, (unsigned *)iota.p, N);
        size_t sort_bytes = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const unsigned *)gdepth.p, (unsigned *)sdepth.p, (const unsigned *)iota.p,
                                                 (unsigned *)order.p, (int)N, 0, depth_bits, st);
        DMG_ALLOC(sorttmp, sort_bytes + 256);
        DMG_HIP(hipcub::DeviceRadixSort::SortPairs(sorttmp.p, sort_bytes, (const unsigned *)gdepth.p, (unsigned *)sdepth.p, (const unsigned *)iota.p,
                                                   (unsigned *)order.p, (int)N, 0, depth_bits, st));
        DMG_ALLOC_VAR(dstart, (size_t)(levels + 1) * 4);
        hipLaunchKernelGGL(k_flow_level_starts, dim3((levels + 256) / 256), dim3(256), 0, st, (const unsigned *)sdepth.p, N, dstart.i(), levels);
        DMG_HIP(hipGetLastError());
        std::vector<int> start(levels + 1, 0);
        DMG_HIP(hipMemcpyAsync(start.data(), dstart.p, (size_t)(levels + 1) * 4, hipMemcpyDeviceToHost, st));
        DMG_HIP(hipStreamSynchronize(st));
        if (timing) {
            fprintf(stderr, "[flow] %d levels\n", levels);
            // level populations: how much of the dependency chain is sparse (counted from the deepest level down)
            const int lim[4] = { 64, 1024, 4096, 16384 };
            int nlev[4] = { 0, 0, 0, 0 }; long long ncell[4] = { 0, 0, 0, 0 };
            for (int q = 0; q < 4; q++)
                for (int l = levels - 1; l >= 0; l--) { const int c = start[l + 1] - start[l]; if (c < lim[q]) { nlev[q]++; ncell[q] += c; } }
            fprintf(stderr, "[flow] levels with < 64 / 1024 / 4096 / 16384 cells: %d / %d / %d / %d (%lld / %lld / %lld / %lld cells of %d reached)\n", nlev[0], nlev[1], nlev[2],
                    nlev[3], ncell[0], ncell[1], ncell[2], ncell[3], start[levels]);
        }
        lap("upload + depth sort");
        const int n_reached = start[levels];
        if (by_levels) {
            for (int l = levels - 1; l >= 0; l--) {
                int cnt = start[l + 1] - start[l];
                if (!cnt) continue;
                hipLaunchKernelGGL(k_flow_level, dim3((cnt + 255) / 256), dim3(256), 0, st, fc, order.i(), start[l], cnt,
                                   (const unsigned short *)mask.p, dcf.i(), down.f(), acc.f());
            }
        } else if (n_reached > 0) {
            int *const dtick_p = tickA.i() + 4;              // (ticket and error words of the one pass: zeroed with the rest)
            DMG_HIP(hipMemsetAsync(acc.p, 0xFF, N * 4, st));                  // FLOW_NOT_READY everywhere
            const int waves = (n_reached + 63) / 64;
            // (Round 5 measured where this kernel's 2.9 ms per config-2 iteration go.  Of ~1830 levels the deepest ~1400 hold fewer than 1024
            // cells each -- the upper reaches of the longest rivers, 0.3 M of 3.1 M cells -- and looked like 1400 x 1.6 us of latency.  They
            // are not: with those levels taken over by ONE workgroup (a thread per cell, the level's sums in LDS, a __syncthreads per level,
            // the additions pre-decoded into 16 bytes per cell: bit-identical, 0.75 ms) this kernel still took 2.28 ms for the rest, and with
            // levels of up to 4096 cells taken over (1790 levels, 1.05 M cells: 2.18 ms on one CU) 1.52 ms for the last 190 levels.  It
            // spends ~0.4 us on a sparse level and is bound by THROUGHPUT, ~1.1 G cells/s, on the dense levels near the mouths, where the
            // cells are: a dozen scattered loads per cell from 512 waves.  The single-workgroup form was dropped.)
            // The persistent grid is SMALL: 128 blocks = 512 waves.  Every wave that holds a ticket polls, and the polls of waves far
            // from the front compete with the loads that matter: with what the chip holds at once (8 192 waves) the sums of a
            // 16384x8192 round took 45 ms, with 512 waves 30 ms (4096x2048: 4.5 -> 3.4 ms; 128 waves are too few: 36 ms).
            // Measured and dropped in the same sitting: a lane that walks on downstream with the sum it has just published
            // (computing the next cell itself, one round of loads per hop instead of a store and a poll) -- as many rounds saved as
            // each round got longer; a round is one coherent load's trip to the memory side, ~1 us, either way.  Nor is there a
            // shorter trip to be had by keeping the whole grid on ONE XCD (workgroups elsewhere leaving at once) and talking through
            // that XCD's L2: workgroup-scope loads (sc0) are served by the CU's own L1 and never see another CU's sum, with or
            // without a `buffer_inv sc0` in front; `buffer_inv sc1` works at 50 ms per solve; agent-scope loads of sums stored at
            // workgroup scope work, and take as long as before (4.0 ms on one XCD against 3.2 ms on eight).
            // (Longer sleeps for waves that have polled in vain for a while -- far from the front -- cost a third at 4096x2048 and
            // gain nothing at 16384x8192, with 128, 256 or 512 blocks: round 5.)
            // (3.40 -> 3.15 ms at 4096x2048; at 16384x8192, where a level holds ten times the cells, its extra loads cost 5 %)
            // (chosen by what was measured to decide it -- how many cells a level holds -- not by the grid's size: 1 700 cells per
            // level at 4096x2048 (3.1 M reached cells, 1 830 levels), 14 700 at 16384x8192 (50 M, 3 400); the line between them at 5 000)
            const int level_pop = (int)(n_reached / std::max(levels, 1));
            const bool sparse_levels = level_pop <= 5000;
            const int look_through = look_env >= 0 ? look_env : (sparse_levels ? 1 : 0);
            // (with a sum asked for only until it has arrived, 4096x2048 takes 256 blocks better than 128: 3.13 -> 2.86 ms; 16384x8192 does not: 32.7 -> 37.0)
            const int blocks = std::min((waves + 3) / 4, nblocks_env > 0 ? nblocks_env : (sparse_levels ? 256 : 128));
            hipLaunchKernelGGL(k_flow_dataflow, dim3(blocks), dim3(256), 0, st, fc, order.i(), n_reached, (const unsigned short *)mask.p, dcf.i(),
                               down.f(), (unsigned *)acc.p, dtick_p, dtick_p + 1, look_through);
            DMG_HIP(hipGetLastError());
            int flow_err = 0;
            DMG_HIP(hipMemcpyAsync(&flow_err, dtick_p + 1, 4, hipMemcpyDeviceToHost, st));
            DMG_HIP(hipStreamSynchronize(st));
            if (flow_err) return dmg_fail(DMG_EHIP, "flow: the upstream sums did not complete (a cell waited for a sum that was never published)");
        }
    }
    hipLaunchKernelGGL(k_flow_pow, dim3(nb), dim3(256), 0, st, gdepth.i(), acc.f(), d_flow, exponent, N);
    DMG_HIP(hipGetLastError());
    if (draw_lakes && nlakes) {
        hipLaunchKernelGGL(k_flow_lakefill, dim3(nb), dim3(256), 0, st, d_h, label.i(), dlevel.f(), d_flow, N);
        DMG_HIP(hipGetLastError());
        DMG_HIP(hipStreamSynchronize(st));
    }
    DMG_HIP(hipStreamSynchronize(st));                   // the temporaries above are released on return
    lap("level sums");
    return DMG_OK;
}

extern "C" {

// The two GLSL passes of FlowFilter::findMagicNumbers (:104-320) on their own: pre-blur, direction codes (keypad
// 1..9, 5 = sink, 0 = sea / unselected) and receiver bitmask, as floats like the reference's textures
int dmg_flow_passes(dmg_ctx *ctx, const dmg_grid *grid, const float *h, const float *sel, float preblur, float *codes_out, float *mask_out)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr && codes_out != nullptr && mask_out != nullptr && preblur >= 0.0f);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const int W = grid->W, H = grid->H;
    const size_t N = (size_t)W * H;
    hipStream_t st = ctx->stream;
    flow_canvas fc;
    fc.c = make_canvas(*grid, topo);
    dmg_grid gm = *grid;                                 // coordsMod (:253-256)
    gm.coords[0] = (float)((double)gm.coords[0] + 1e-3);
    gm.coords[1] = (float)((double)gm.coords[1] - 1e-3);
    fc.cm = make_canvas(gm, dmg_topology_of(gm));
    fc.wrap = fabs(grid->coords[3] - grid->coords[2]) > 2 * M_PI - 1e-4;
    std::vector<float> dxr(H), dyr(H);
    for (int y = 0; y < H; y++) dmg_row_pixelsize(gm, y, &dxr[y], &dyr[y]);
    dev_buf ddx, ddy, hb, tmp, dsel, code, mask;
    if ((rc = upload_vec(ctx, ddx, dxr)) || (rc = upload_vec(ctx, ddy, dyr))) return rc;
    DMG_ALLOC(hb, N * 4); DMG_ALLOC(tmp, N * 4); DMG_ALLOC(code, N); DMG_ALLOC(mask, N * 2);
    DMG_HIP(hipMemcpyAsync(hb.p, h, N * 4, hipMemcpyHostToDevice, st));
    if (sel) { DMG_ALLOC(dsel, N * 4); DMG_HIP(hipMemcpyAsync(dsel.p, sel, N * 4, hipMemcpyHostToDevice, st)); }
    if (preblur > 0.0f && (rc = blur_device(ctx, *grid, topo, hb.f(), tmp.f(), preblur))) return rc;
    dim3 g2((W + 255) / 256, H);
    hipLaunchKernelGGL(k_flow_direction, g2, dim3(256), 0, st, fc, hb.f(), sel ? dsel.f() : nullptr, ddx.f(), ddy.f(), (unsigned char *)code.p);
    hipLaunchKernelGGL(k_flow_mask, g2, dim3(256), 0, st, fc, hb.f(), sel ? dsel.f() : nullptr, (const unsigned char *)code.p, (unsigned short *)mask.p);
    DMG_HIP(hipGetLastError());
    std::vector<unsigned char> hc(N);
    std::vector<unsigned short> hm(N);
    DMG_HIP(hipMemcpyAsync(hc.data(), code.p, N, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipMemcpyAsync(hm.data(), mask.p, N * 2, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    // the reference's texture values: -1 off the land, else the bit sum on top of its 0.5 seed (:269-303)
    for (size_t i = 0; i < N; i++) { codes_out[i] = (float)hc[i]; mask_out[i] = hm[i] == 0xFFFFu ? -1.0f : (float)hm[i] + 0.5f; }
    return DMG_OK;
}

// FlowFilter::run (:719-786): flow_out[i] = pow(upstream area, exponent), -1 off the drained land; lakes != 0 also
// draws the lakes (lakeflag, :660-699)
int dmg_flow_accumulate(dmg_ctx *ctx, const dmg_grid *grid, const float *h, const float *sel, float preblur,
                        float exponent, int lakes, float *flow_out)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr && flow_out != nullptr && preblur >= 0.0f);
    dmg_topology topo;
    int rc = dmg_check_grid(grid, &topo);
    if (rc) return rc;
    DMG_HIP(hipSetDevice(ctx->device));
    const size_t N = (size_t)grid->W * grid->H;
    dev_buf dh, dsel, dflow;
    DMG_ALLOC(dh, N * 4); DMG_ALLOC(dflow, N * 4);
    DMG_HIP(hipMemcpyAsync(dh.p, h, N * 4, hipMemcpyHostToDevice, ctx->stream));
    if (sel) {
        DMG_ALLOC(dsel, N * 4);
        DMG_HIP(hipMemcpyAsync(dsel.p, sel, N * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = flow_device(ctx, *grid, topo, dh.f(), sel ? dsel.f() : nullptr, preblur, exponent, dflow.f(), lakes))) return rc;
    DMG_HIP(hipMemcpyAsync(flow_out, dflow.p, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMG_HIP(hipStreamSynchronize(ctx->stream));
    return DMG_OK;
}

}  // extern "C"
