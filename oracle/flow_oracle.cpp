// flow_oracle.cpp -- CPU restatement of FlowFilter::run (reference src/filter/FlowFilter.cpp:719-786), lakeflag = 0.
// TEST INFRASTRUCTURE ONLY (parity checker), part of liboracle.so.
//
// Follows the reference stage by stage (findMagicNumbers, indexLakes, assignLakeIds, findAllConnections,
// solvingConnections, calculateflow) on the host.  Three places of the reference are order-dependent on container
// internals and thread timing and therefore not reproducible even by the reference itself:
//   * passes of one lake live in a std::set ordered by height ONLY (FlowFilter.h:62-63): equal heights collapse and
//     the survivor depends on std::unordered_map<float,pass> iteration order (:449,:493);
//   * the candidate set of solvingConnections has the same comparator (:534);
//   * `lakes` is filled in thread-completion order (:353-355) and iterated in that order.
// Here every such choice uses the canonical total order (height, from, tolocation) and lakes are visited in
// ascending cell index; the HIP backend implements the same rule.
// Replicated literally (flag `fix_mouth_filter` = 0, the default): `if (Nthbit(c.from,10)) continue;` (:552) tests bit 10
// of the neighbouring lake's CELL INDEX, not of its data word, so a river-mouth lake's first candidate skips (and
// loses) every pass whose `from` sink has bit 512 set in its index.  Not replicated: the float-keyed pass map (:449)
// aliases lake ids above 2^24 (grids beyond 16.7 M cells only) in DFS order.
// Host arithmetic: with the reference's includes (<cmath> through glm and the std headers, never <math.h>) the
// unqualified `cos(geoy)` (:613) and `pow(sum,exponent)` (:620) resolve to the C `double` functions under g++
// (checked here with the same header set: sizeof(pow(float,float)) == 8), so both are evaluated in double and rounded
// to float once.
#include "dmg_oracle.h"
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {
struct Pass { float h; int from; int tolocation; };
struct PassLess {
    bool operator()(const Pass &a, const Pass &b) const
    { return std::tie(a.h, a.from, a.tolocation) < std::tie(b.h, b.from, b.tolocation); }
};
inline bool nthbit(int num, int n) { return num & (1 << (n - 1)); }

struct Topo { int W, H; bool wrap; };
// FlowFilter::neighbours (:39-75): cells selected by the bits of `dat`, keypad order 1,2,3,4,6,7,8,9
inline int neighbours(const Topo &t, int pos, int dat, int out[8])
{
    static const int bit[8] = { 1, 2, 3, 4, 6, 7, 8, 9 };
    static const int dx[8] = { -1, 0, 1, -1, 1, -1, 0, 1 }, dy[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
    int n = 0;
    for (int k = 0; k < 8; k++) {
        if (!nthbit(dat, bit[k])) continue;
        int x = pos % t.W + dx[k], y = pos / t.W + dy[k];
        if (t.wrap) x = (x + t.W) % t.W;
        else if (x >= t.W || x < 0) continue;
        if (y >= t.H || y < 0) continue;
        out[n++] = y * t.W + x;
    }
    return n;
}
}  // namespace

static void flow_accumulate(const orc_grid *g, const float *h, const float *sel_in, float preblur, float exponent,
                            int lakeflag, float *flow, float *codes_out, float *mask_out, int *label_out, int fix_mouth_filter = 0)
{
    const int W = g->W, H = g->H;
    const size_t N = (size_t)W * H;
    Topo topo{ W, H, std::fabs(g->coords[3] - g->coords[2]) > 2 * M_PI - 1e-4 };      // :47
    std::vector<float> sel(N, 1.0f);                                                  // Project.cpp:102-104
    if (sel_in) sel.assign(sel_in, sel_in + N);

    // findMagicNumbers: blur the terrain (:90-103), direction + bitmask passes, restore the terrain (:314)
    std::vector<float> hb(h, h + N), scratch(N), codes(N), data(N);
    if (preblur > 0.0f) orc_blur(g, hb.data(), scratch.data(), preblur);
    orc_flow_direction(g, hb.data(), sel.data(), codes.data());
    orc_flow_mask(g, hb.data(), sel.data(), codes.data(), data.data());

    // indexLakes (:344-358): cells whose data word has bit 5 (sink)
    std::vector<int> lakes;
    for (size_t i = 0; i < N; i++)
        if (data[i] >= 0 && nthbit((int)data[i], 5)) lakes.push_back((int)i);

    // assignLakeIds (:360-398): label every cell upstream of a sink with that sink's cell index
    std::vector<int> label(N, -1);
    std::vector<int> stack;
    int nb[8];
    for (int lake : lakes) {
        stack.assign(1, lake);
        while (!stack.empty()) {
            int s = stack.back(); stack.pop_back();
            label[s] = lake;
            int n = neighbours(topo, s, (int)data[s], nb);
            for (int k = 0; k < n; k++) stack.push_back(nb[k]);
        }
    }

    // findAllConnections (:400-531): adjacency pass on float-encoded ids, then the lowest pass towards each
    // neighbouring lake
    std::vector<float> lakeid(N), passbits(N);
    for (size_t i = 0; i < N; i++) {
        if (label[i] < 0) lakeid[i] = -1.0f;
        else { int bits = 1073741824 + label[i]; std::memcpy(&lakeid[i], &bits, 4); }   // :373
    }
    orc_lake_adjacency(g, lakeid.data(), passbits.data());
    std::map<int, std::set<Pass, PassLess>> passes;       // lake -> passes INTO it, from neighbouring lakes
    for (int lake : lakes) {
        std::map<int, Pass> best;                          // neighbouring lake -> lowest pass
        stack.assign(1, lake);
        while (!stack.empty()) {
            int s = stack.back(); stack.pop_back();
            if (passbits[s] > 0) {
                float minpass = INFINITY; int nlake = -1;
                int n = neighbours(topo, s, (int)passbits[s], nb);
                for (int k = 0; k < n; k++) {
                    float bd = h[nb[k]];
                    if (label[nb[k]] != lake && bd > 0 && bd < minpass) { minpass = bd; nlake = nb[k]; }
                }
                if (nlake >= 0) {
                    int lid = label[nlake];
                    if (lid >= 0 && !nthbit((int)data[lid], 10)) {           // not towards a river-mouth lake (:484)
                        Pass p{ std::fmax(minpass, h[s]), lid, s };
                        auto it = best.find(lid);
                        if (it == best.end() || PassLess()(p, it->second)) best[lid] = p;
                    }
                }
            }
            int n = neighbours(topo, s, (int)data[s], nb);
            for (int k = 0; k < n; k++) stack.push_back(nb[k]);
        }
        auto &set = passes[lake];
        for (auto &kv : best) set.insert(kv.second);
    }

    // solvingConnections (:533-595): grow from the river-mouth lakes, always taking the lowest open pass
    std::set<Pass, PassLess> candidates;
    std::set<int> placed;
    std::unordered_map<int, Pass> connections;             // tolocation -> pass (:588)
    auto push_next = [&](int lake, bool initial = false) {
        auto &set = passes[lake];
        while (!set.empty()) {
            Pass c = *set.begin();
            set.erase(set.begin());
            if (placed.count(c.from)) continue;
            if (initial && !fix_mouth_filter && nthbit(c.from, 10)) continue;     // :552, bit 10 of a cell index
            candidates.insert(c);
            break;
        }
    };
    for (int lake : lakes) {
        if (!nthbit((int)data[lake], 10)) continue;        // only river mouths (:543)
        placed.insert(lake);
    }
    for (int lake : lakes)
        if (nthbit((int)data[lake], 10)) push_next(lake, true);
    while (!candidates.empty()) {
        Pass p = *candidates.begin();
        candidates.erase(candidates.begin());
        int to_lake = label[p.tolocation];
        if (placed.count(p.from)) { push_next(to_lake); continue; }
        placed.insert(p.from);
        connections[p.tolocation] = p;                     // a later pass at the same cell replaces an earlier one
        push_next(p.from);
        push_next(to_lake);
    }

    // calculateflow (:597-717): upstream area, summed in the reference's order (neighbours, then the connection)
    std::vector<float> out(N, -1.0f);
    float pixelwidthx = (float)(g->circumference * (g->coords[3] - g->coords[2]) / (2 * M_PI) / W);
    float pixelwidthy = (float)(g->circumference * (g->coords[1] - g->coords[0]) / (2 * M_PI) / H);
    auto own = [&](int p) {
        float y = (float)(p / W) / H;
        float geoy = (y * (g->coords[1] - g->coords[0]) + g->coords[0]);
        return (float)(pixelwidthy * pixelwidthx * std::cos((double)geoy) * 1e-5);          // double cos, see the header
    };
    struct Frame { int p; int child; float sum; int kids[9]; int nk; };
    std::vector<Frame> st;
    for (int lake : lakes) {
        if (!nthbit((int)data[lake], 10)) continue;
        auto open = [&](int p) {
            Frame f; f.p = p; f.child = 0; f.sum = own(p);
            f.nk = neighbours(topo, p, (int)data[p], f.kids);
            auto it = connections.find(p);
            if (it != connections.end()) f.kids[f.nk++] = it->second.from;
            st.push_back(f);
        };
        open(lake);
        float ret = 0;
        bool returning = false;
        while (!st.empty()) {
            Frame &f = st.back();
            if (returning) { f.sum += ret; returning = false; }
            if (f.child < f.nk) { int c = f.kids[f.child++]; open(c); continue; }
            out[f.p] = (float)std::pow((double)f.sum, (double)exponent);   // lakeID[p] = pow(sum, exponent) (:620): C double pow
            ret = f.sum; returning = true;
            st.pop_back();
        }
    }
    // lake drawing (:660-699): walking the lake tree from every river mouth, a lake's water level is the highest
    // pass on the way down to it; cells of the lake's catchment at or below that level are marked 0
    if (lakeflag) {
        struct Job { int lake; float waterheight; };
        std::vector<Job> jobs;
        for (int lake : lakes)
            if (nthbit((int)data[lake], 10)) jobs.push_back({ lake, 0.0f });
        while (!jobs.empty()) {
            Job j = jobs.back(); jobs.pop_back();
            stack.assign(1, j.lake);
            while (!stack.empty()) {
                int p = stack.back(); stack.pop_back();
                if (h[p] <= j.waterheight) out[p] = 0;
                int n = neighbours(topo, p, (int)data[p], nb);
                for (int k = 0; k < n; k++) stack.push_back(nb[k]);
                auto it = connections.find(p);
                if (it != connections.end()) {
                    const Pass &c = it->second;
                    jobs.push_back({ c.from, j.waterheight > c.h ? j.waterheight : c.h });
                }
            }
        }
    }
    std::memcpy(flow, out.data(), N * sizeof(float));
    if (codes_out) std::memcpy(codes_out, codes.data(), N * sizeof(float));
    if (mask_out) std::memcpy(mask_out, data.data(), N * sizeof(float));
    if (label_out) std::memcpy(label_out, label.data(), N * sizeof(int));
}

extern "C" void orc_flow_accumulate(const orc_grid *g, const float *h, const float *sel, float preblur, float exponent,
                                    float *flow, float *codes, float *mask, int *label)
{
    flow_accumulate(g, h, sel, preblur, exponent, 0, flow, codes, mask, label);
}
extern "C" void orc_flow_accumulate_flags(const orc_grid *g, const float *h, const float *sel, float preblur, float exponent,
                                          int lakeflag, int fix_mouth_filter, float *flow)
{
    flow_accumulate(g, h, sel, preblur, exponent, lakeflag, flow, nullptr, nullptr, nullptr, fix_mouth_filter);
}
extern "C" void orc_flow_accumulate_lakes(const orc_grid *g, const float *h, const float *sel, float preblur, float exponent,
                                          float *flow)
{
    flow_accumulate(g, h, sel, preblur, exponent, 1, flow, nullptr, nullptr, nullptr);
}
