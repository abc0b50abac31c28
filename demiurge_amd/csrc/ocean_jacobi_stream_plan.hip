// The streaming Jacobi kernel's host side: the cut of a launch into units, their pricing, the simulated dispatch and the placement
// (DESIGN.md 4.1b).  No device code: dmg_stream_units_probe runs all of it without a GPU (tests/test_stream_cut.py).
#include "ocean_stream_dev.h"
#include "ocean_stream_plan.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

int st_unit_rows()                                 // DMG_ST_ROWS=<n>: fixed unit height (experiments); default 0 = chosen per launch geometry
{
    static const int rows = [] {
        const char *e = dmg_hook("DMG_ST_ROWS");
        int v = e ? atoi(e) : 0;
        return v >= 1 && v <= ST_MAXROWS ? v : 0;
    }();
    return rows;
}

namespace {
struct st_unit { int tx, r0, r1; float cost; int flip = 0;         // flip: the unit streams southwards (k_jacobi_stream's second view)
                 int rlo = -1;                                     // >= 0: a pair of twin units (k_jacobi_stream_twin) -- rows [r0, r1) northwards and [rlo, r0) southwards from the seam r0
                 float cost_n = 0.0f, cost_s = 0.0f;               // twin pair: the northward and the southward unit's cost (cost = 16 x the dearer)
                 int south_first = 0;                              // twin pair: the southward unit is the workgroup's first wave (st_place_twins)
                 int dummy = 0;                                    // a workgroup that ends at once (keeps its successors on the CUs st_place_twins meant them for)
                 int first() const { return rlo >= 0 ? rlo : r0; } };
}

// The planner's independent pieces -- candidate cuts, the stretches of a balanced cut -- side by side on the host's cores
// (DMG_ST_PLAN_THREADS: experiments; 1 = in turn).  Whatever the order they finish in, the results are the ones a single thread computes.
template <class F>
static void st_parallel_for(size_t n, F &&f)
{
    static const int threads_env = [] { const char *e = dmg_hook("DMG_ST_PLAN_THREADS"); return e ? atoi(e) : 0; }();
    const int nthreads = (int)std::max<size_t>(1, std::min<size_t>({ (size_t)(threads_env > 0 ? threads_env : 16), (size_t)std::thread::hardware_concurrency(), n }));
    if (nthreads <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{ 0 };
    auto work = [&]() { for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i); };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
}

// What the steps of a unit are made of, as the kernel will run them (st_step): a step whose ten rows in flight are all open
// water runs its eight levels as one block without a test; one with coast rows but no row that is land everywhere runs every
// row through the coast formula; every other step -- the pipeline's fill, a land row in flight -- takes the tested path, level by
// level and row by row.  cls: the strip's row classes (null: all mixed), rows outside [0, L) and outside the grid count as mixed
// as in the kernel; flip: the unit runs on the mirrored view (rows r0, r1 are that view's); twin: its window starts on r0.
// Round 6: a row whose divisors have no exact two-operation pair (div2.hip: 8 of the 8192 rows of the headline grid) puts the five
// steps it is in flight for on the tested path like a land row does (st_step's `special`) -- at twice an open-water step each.  On
// the full grid that is nothing; in a one-round launch the unit that holds such a row was the one every other unit waited for
// (profiles/r06_band_units.txt: the 36-row units around local row 512 of bands 1 and 6 of the 8-way split took 55 us, the rest 45).
// Priced at three open-water steps (DMG_ST_WBAD; per-unit timings fit 2.0, a same-box sweep of the weight was flat from 3 up:
// profiles/r06_band_probe.log) the unit that holds such a row is cut shorter: the slowest band of the 8-way split 58.1 -> 47.8 us.
st_features st_unit_features(const unsigned char *cls, int L, int y_first, int H, int r0, int r1, bool flip, bool twin, const unsigned char *bad)
{
    st_features f;
    const int e0 = twin ? r0 : r0 - TB_T, nsteps = (((r1 - r0) + (twin ? TB_T : 2 * TB_T) + 3) / 4) * 2;
    const int yf = flip ? H - y_first - L : y_first;
    auto cl = [&](int lu) -> int {
        if (lu < e0) return 1;                              // rows before the window: "land" (st_state::landbits)
        const int yg = yf + lu;
        if (!cls || lu < 0 || lu >= L || yg < 0 || yg >= H) return 2;
        return cls[flip ? L - 1 - lu : lu];
    };
    auto bd = [&](int lu) -> bool { return bad && lu >= e0 && lu >= 0 && lu < L && bad[flip ? L - 1 - lu : lu]; };
    for (int t = 0; t < nsteps; t++) {
        const int newest = e0 + 2 * t + 1;
        bool land = false, mix = false, special = false;
        for (int k = 0; k < 10; k++) { const int c = cl(newest - k); land |= c == 1; mix |= c == 2; special |= bd(newest - k); }
        const bool filled = t >= (twin ? TB_T / 2 : TB_T);
        if (filled && !land && !mix && !special) f.fast++;
        else if (filled && !land && !special) f.coast++;
        else if (filled && !land) f.bad++;
        else {
            if (filled) f.tested++;
            for (int S = 1; S <= 8; S++) {
                if (twin ? S > 2 * t + 1 : S > t) continue;
                for (int k = S - 1; k <= S; k++) {
                    const int c = cl(newest - 1 - k);
                    if (filled) (c == 0 ? f.t_open : (c == 2 ? f.t_mixed : f.t_land))++;
                    else (c == 0 ? f.f_open : (c == 2 ? f.f_mixed : f.f_land))++;
                }
            }
        }
    }
    return f;
}

// A unit's predicted time in "open-water steps", from what its steps are made of.  The weights are the marginal costs that
// per-unit timings of a 16384 x 1024 band give when regressed on these counts (DMG_ST_PROF, six launches of two cuts, younger
// and older wave of a SIMD apart; profiles/r04_twin_units.txt): a step through the coast formula 1.6 .. 1.7 open-water steps;
// a tested step (a land row in flight) per row and level it visits 0.05 for an open row, 0.14 for a coast row and 0.11 for a
// LAND row -- a land row computes nothing, but a level visit on the tested path costs its tests and branches whatever its
// rows are, which is what round 3's step clocks showed for skipped levels; the fill steps 0.37 + the same per row and level
// (step clocks: 1.0 + 0.12 us per level-row against 2.7 us for a full step).  r0, r1 in the band's own numbering; flip: the
// unit streams southwards; twin: its window starts on its first row (no halo there).
struct st_weights {
    float coast, t_open, t_mixed, t_land, bad;
    static const st_weights &get()
    {
        static const st_weights w = [] {
            auto env = [](const char *name, float dflt) { const char *e = dmg_hook(name); return e ? (float)atof(e) : dflt; };
            return st_weights{ env("DMG_ST_WCOAST", 1.7f), env("DMG_ST_WTOPEN", 0.05f), env("DMG_ST_WTMIXED", 0.14f), env("DMG_ST_WTLAND", 0.11f), env("DMG_ST_WBAD", 3.0f) };
        }();
        return w;
    }
};
static float st_exact_cost(const dmg_ocean *o, int tx, int r0, int r1, bool flip, bool twin)
{
    const st_weights &w = st_weights::get();
    const unsigned char *cls = o->rowclass_host.empty() ? nullptr : o->rowclass_host.data() + (size_t)tx * o->L;
    const st_features f = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, flip ? o->L - r1 : r0, flip ? o->L - r0 : r1, flip, twin,
                                           o->rowbad_host.empty() ? nullptr : o->rowbad_host.data());
    const float fill_steps = twin ? 4.0f : 8.0f;
    return 0.7f + (float)f.fast + w.coast * (float)f.coast + w.bad * (float)f.bad + 0.37f * fill_steps + w.t_open * (float)(f.t_open + f.f_open) + w.t_mixed * (float)(f.t_mixed + f.f_mixed) +
           w.t_land * (float)(f.t_land + f.f_land);
}

// what the dispatch simulation charges a unit of rows [r0, r1) of a strip (cls: the strip's row classes, or null): its
// level-rows, a coast row counted 1.4 times an open one and a land row a tenth (what the kernel spends on them), plus what a
// unit costs before and after its rows
static float st_unit_cost(const unsigned char *cls, int L, int r0, int r1, int variant, int twin = 0)    // twin 1: no halo below r0 (a unit that starts on a seam), 2: none above r1
{
    static const float w_mixed = [] { const char *e = dmg_hook("DMG_ST_WMIXED"); return e ? (float)atof(e) : 1.4f; }();    // experiments: the cost model's weights (1.4: best of a sweep 1.0 .. 1.6 on the survey terrain; the per-unit timings fit 1.23 .. 1.40)
    static const float w_land = [] { const char *e = dmg_hook("DMG_ST_WLAND"); return e ? (float)atof(e) : 0.1f; }();
    float cost = 0.0f;
    for (int r = twin == 1 ? r0 : r0 - TB_T; r < (twin == 2 ? r1 : r1 + TB_T); r++) {
        const int c = (cls && r >= 0 && r < L) ? cls[r] : 2;
        const int below = twin == 1 ? TB_T : r - (r0 - TB_T), above = twin == 2 ? TB_T : (r1 + TB_T - 1) - r;
        const int depth = std::min(std::min(below, above), TB_T);   // levels that still need this row
        cost += (c == 1 ? w_land : (c == 2 ? w_mixed : 1.0f)) * (float)depth;
    }
    // + what a unit costs before and after its rows: dispatch, the class and coefficient tables, the first loads' round
    // trip, the last stores.  Per-unit timings (DMG_ST_PROF) fit duration = 0.152 us x cost + 22 us: ~140 cost units,
    // not the 4 the first version assumed -- which made the simulation prefer many short units
    static const float unit_cost = [] { const char *e = dmg_hook("DMG_ST_UNITCOST"); return e ? (float)atof(e) : 140.0f; }();
    // an open-water unit (st_open_unit: no row of its window is anything but open water) spends ~440 instead of ~590
    // instructions per step
    static const float w_openunit = [] { const char *e = dmg_hook("DMG_ST_WOPEN"); return e ? (float)atof(e) : 1.0f; }();     // (experiments; 0.78 is what their instruction count suggests and what made the cut worse)
    if (cls && !variant && w_openunit != 1.0f) {
        bool all_open = r0 - TB_T >= 0 && r1 + TB_T + 6 <= L;
        for (int r = r0 - TB_T; all_open && r < r1 + TB_T + 6; r++) all_open = cls[r] == 0;
        if (all_open) cost *= w_openunit;
    }
    return cost + unit_cost;
}

// Units of a launch over local rows [lo, hi) for one unit height: every strip cut into segments of `seg` rows, each trimmed
// to its first / last row that is not land everywhere; empty ones are dropped.  cost ~ the unit's level-rows, a coast row
// counted 1.4 times an open one and a land row a tenth (what the kernel spends on them).
static void st_cut(const dmg_ocean *o, int variant, int lo, int hi, int seg, int part, int zone_lo, int zone_hi, std::vector<st_unit> &units,
                   int seg2 = 0, int split = 0, int edge_seg = 0)
{
    const int ow = TB_OW;
    const std::vector<unsigned char> &table = o->rowclass_host;
    const int tiles_x = (o->grid.W + ow - 1) / ow, L = o->L;
    const bool classes = !table.empty();
    units.clear();
    // rows [lo, split) in segments of `seg` rows, the rest in (shorter) segments of `seg2` rows: the short units fill
    // the last round of the dispatch
    if (!seg2) split = hi;
    // edge_seg > 0 (the halves of a split launch): the send zones [lo, zone_lo) and [zone_hi, hi) are cut into units of edge_seg
    // rows of their own (4: measured 4 .. 32) and the rows between them as the whole launch cuts them, anchored at zone_lo.  An edge unit of 8 rows
    // is done in ~25 us, one of 28 in ~45: the exchange behind the edge units starts that much earlier, and what it can hide
    // behind is the interior units of one launch (~57 us on a 1024-row band) -- see profiles/r03_fake_link.log
    const bool zones = edge_seg > 0 && zone_lo >= lo && zone_hi <= hi && zone_lo <= zone_hi;
    const int mid_lo = zones ? zone_lo : lo, mid_hi = zones ? zone_hi : hi;
    // every other segment streams southwards (k_jacobi_stream): neighbouring units then start, or end, on the rows they share
    static const int flip_env = [] { const char *e = dmg_hook("DMG_ST_FLIP"); return e ? atoi(e) : -1; }();
    const bool flipping = variant == 0 && (flip_env >= 0 ? flip_env != 0 : ST_FLIP_DEFAULT != 0);
    int segi = 0;
    for (int s0 = lo; s0 < hi; segi++) {
        int s1;
        if (s0 < mid_lo) s1 = std::min(s0 + edge_seg, mid_lo);
        else if (s0 >= mid_hi) s1 = std::min(s0 + edge_seg, hi);
        else {
            const int h = s0 < split ? seg : seg2;
            s1 = std::min(std::min(s0 + h, s0 < split ? split : hi), mid_hi);
            if (s1 <= s0) s1 = std::min(s0 + seg, mid_hi);                  // (a split that lies below the middle zone)
        }
        const int s0_this = s0;
        s0 = s1;
        {
            const int s0 = s0_this;
        for (int tx = 0; tx < tiles_x; tx++) {
            int r0 = s0, r1 = s1;
            const unsigned char *cls = classes ? table.data() + (size_t)tx * L : nullptr;
            if (cls) {
                while (r0 < r1 && cls[r0] == 1) r0++;
                while (r1 > r0 && cls[r1 - 1] == 1) r1--;
            }
            if (r0 >= r1) continue;
            if (part) {
                const bool edge = r0 < zone_lo || r1 > zone_hi;
                if (edge != (part == 1)) continue;
            }
            const float cost = st_unit_cost(cls, L, r0, r1, variant), unit_cost = 0.0f;
            units.push_back(st_unit{ tx, r0, r1, cost + unit_cost, (flipping && (segi & 1)) ? 1 : 0 });
        }
        }
    }
}

// A cut for launches of about one round (round 4).  A band of a thousand rows has about as many units as the chip has wave
// slots, so the launch lasts as long as its SLOWEST unit -- and with one height for every strip the slowest are the coast
// units (a coast row costs 1.4 of an open one): of a 16384 x 1024 band's 1831 units the median takes 59 us, the launch 65
// (DMG_ST_PROF), while 11 % of the slots stay empty.  Here the middle zone of the launch is cut strip by strip into `want`
// units of about EQUAL COST: every stretch of water (land runs of 24 rows and more are left out) starts as one unit, and the
// stretch whose units are the most expensive gets one more until `want` are dealt out; a stretch's units then split its rows
// at equal cumulative cost.  The send zones of a split launch keep their short units (st_cut).  Which cut a launch takes
// is still decided by the simulated dispatch (st_units): this one is a candidate at one, two and three rounds' worth of units.
// twin: the stretches are cut into PAIRS of units that start on a common seam (k_jacobi_stream_twin; `want` still counts units)
static void st_cut_balanced(const dmg_ocean *o, int lo, int hi, int want, int part, int zone_lo, int zone_hi, std::vector<st_unit> &units, int edge_seg = 0,
                            bool twin = false)
{
    const std::vector<unsigned char> &table = o->rowclass_host;
    const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW, L = o->L;
    const bool classes = !table.empty();
    static const int flip_env = [] { const char *e = dmg_hook("DMG_ST_FLIP"); return e ? atoi(e) : -1; }();
    const bool flipping = flip_env >= 0 ? flip_env != 0 : ST_FLIP_DEFAULT != 0;
    const bool zones = edge_seg > 0 && zone_lo >= lo && zone_hi <= hi && zone_lo <= zone_hi;
    const int mid_lo = zones ? zone_lo : lo, mid_hi = zones ? zone_hi : hi;
    units.clear();
    int n_all = 0;                                           // units of the whole launch so far (a part keeps only its own, but both parts cut alike)
    auto emit = [&](int tx, const unsigned char *cls, int r0, int r1, int flip) {
        if (cls) {
            while (r0 < r1 && cls[r0] == 1) r0++;
            while (r1 > r0 && cls[r1 - 1] == 1) r1--;
        }
        if (r0 >= r1) return;
        n_all++;
        if (part) {
            const bool edge = r0 < zone_lo || r1 > zone_hi;
            if (edge != (part == 1)) return;
        }
        units.push_back(st_unit{ tx, r0, r1, st_unit_cost(cls, L, r0, r1, 0), flip });
    };
    // a pair of twin units around the seam: rows [rlo, seam) southwards, [seam, r1) northwards; only their far ends are trimmed
    auto emit_twin = [&](int tx, const unsigned char *cls, int rlo, int seam, int r1) {
        if (cls) {
            while (rlo < seam && cls[rlo] == 1) rlo++;
            while (r1 > seam && cls[r1 - 1] == 1) r1--;
        }
        if (rlo >= r1) return;
        n_all += 2;
        if (part) {
            const bool edge = rlo < zone_lo || r1 > zone_hi;
            if (edge != (part == 1)) return;
        }
        const float cn = st_exact_cost(o, tx, seam, r1, false, true), cs = st_exact_cost(o, tx, rlo, seam, true, true);
        st_unit u{ tx, seam, r1, 16.0f * std::max(cn, cs), 0 };
        u.rlo = rlo; u.cost_n = cn; u.cost_s = cs;
        units.push_back(u);
    };
    // the send zones: short units of their own, as in st_cut -- in a twin cut as pairs of such units around a seam (a unit of four
    // rows then runs six steps instead of ten: the edge half of a split launch is what the exchange waits for).  DMG_ST_TWIN_EDGES=0:
    // plain short units in a twin cut's zones as well
    const bool twin_edges = [] { const char *e = dmg_hook("DMG_ST_TWIN_EDGES"); return !e || atoi(e) != 0; }();
    if (zones)
        for (int tx = 0; tx < tiles_x; tx++) {
            const unsigned char *cls = classes ? table.data() + (size_t)tx * L : nullptr;
            if (twin && twin_edges) {
                auto zone_pairs = [&](int z0, int z1) {
                    for (int s0 = z0; s0 < z1; s0 += 2 * edge_seg) {
                        const int s2 = std::min(s0 + 2 * edge_seg, z1), seam = std::min(s0 + edge_seg, s2);
                        emit_twin(tx, cls, s0, seam, s2);
                    }
                };
                zone_pairs(lo, mid_lo);
                zone_pairs(mid_hi, hi);
                continue;
            }
            int k = 0;
            for (int s0 = lo; s0 < mid_lo; s0 += edge_seg, k++) emit(tx, cls, s0, std::min(s0 + edge_seg, mid_lo), flipping && (k & 1));
            for (int s0 = mid_hi; s0 < hi; s0 += edge_seg, k++) emit(tx, cls, s0, std::min(s0 + edge_seg, hi), flipping && (k & 1));
        }
    // stretches of water in the middle zone
    struct stretch { int tx, r0, r1, k; double cost; };
    std::vector<stretch> st;
    static const double w_mixed = [] { const char *e = dmg_hook("DMG_ST_WMIXED"); return e ? atof(e) : 1.4; }();
    // twin units: what a row adds to its unit is what the STEP it arrives in costs (st_exact_cost's weights: its ten rows in flight
    // decide the step's form), so that a stretch's weight, the number of units it is given and its first cut are priced the way
    // the finished units will be.  (Priced row by row -- a coast row 1.4, a land row 0.1 -- the stretches along the coasts got
    // too few units: forcing the coast weight to 2.4 brought the 8-way band from 52.9 to 48.1 us, profiles/r04_twin_units.txt.)
    std::vector<float> step_w;
    if (twin && classes) {
        const st_weights &w = st_weights::get();
        step_w.assign((size_t)tiles_x * L, 1.0f);
        for (int tx = 0; tx < tiles_x; tx++) {
            const unsigned char *cls = table.data() + (size_t)tx * L;
            auto cl = [&](int r) { return r >= 0 && r < L ? (int)cls[r] : 2; };
            const unsigned char *bad = o->rowbad_host.empty() ? nullptr : o->rowbad_host.data();
            for (int r = std::max(lo - 1, 0); r < std::min(hi + 1, L); r++) {
                bool land = false, mix = false, special = false;
                for (int k = 0; k < 10; k++) { const int c = cl(r - k); land |= c == 1; mix |= c == 2; special |= bad && r - k >= 0 && bad[r - k]; }
                float v;
                if (land) {
                    v = 0.0f;
                    for (int S = 1; S <= 8; S++)
                        for (int k = S; k <= S + 1; k++) { const int c = cl(r - k); v += c == 0 ? w.t_open : (c == 2 ? w.t_mixed : w.t_land); }
                } else v = special ? w.bad : (mix ? w.coast : 1.0f);
                step_w[(size_t)tx * L + r] = v;
            }
        }
    }
    auto roww = [&](const unsigned char *cls, int r) -> double {
        if (!step_w.empty()) return step_w[(size_t)(cls - table.data()) + r];
        const int c = cls ? cls[r] : 0;
        return c == 1 ? 0.1 : (c == 2 ? w_mixed : 1.0);
    };
    for (int tx = 0; tx < tiles_x; tx++) {
        const unsigned char *cls = classes ? table.data() + (size_t)tx * L : nullptr;
        int r = mid_lo;
        while (r < mid_hi) {
            while (r < mid_hi && cls && cls[r] == 1) r++;
            if (r >= mid_hi) break;
            const int s0 = r;
            int last = r;
            double c = 0.0;
            while (r < mid_hi) {
                if (cls && cls[r] == 1) {
                    int e = r;
                    while (e < mid_hi && cls[e] == 1) e++;
                    if (e - r >= 24 || e == mid_hi) break;
                }
                c += roww(cls, r);
                if (!cls || cls[r] != 1) last = r;
                r++;
            }
            if (twin) st.push_back(stretch{ tx, s0, last + 1, 2 * std::max(1, (last + 1 - s0 + 2 * ST_TWIN_MAXROWS - 1) / (2 * ST_TWIN_MAXROWS)), c });
            else st.push_back(stretch{ tx, s0, last + 1, std::max(1, (last + 1 - s0 + ST_MAXROWS - 1) / ST_MAXROWS), c });
            r = std::max(r, last + 1);
        }
    }
    if (st.empty()) return;
    // a unit of a stretch costs about (its share of the rows) x 8 levels + the fill of its window + the fixed part
    const double fixed = 140.0 + (twin ? 8.0 : 16.0) * 4.5;
    const int inc = twin ? 2 : 1;
    auto per_unit = [&](const stretch &x) { return x.cost * 8.0 / x.k + fixed; };
    int have = 0;
    for (const stretch &x : st) have += x.k;
    if (!twin) want -= n_all;                                // (the send zones' units are part of the round; the twin units of a split launch's
                                                             // interior are a launch of their own, behind the zones' plain units: a round to themselves)
    {
        auto worse = [&](int a, int b) { return per_unit(st[a]) < per_unit(st[b]); };
        std::vector<int> heap(st.size());
        for (size_t i = 0; i < st.size(); i++) heap[i] = (int)i;
        std::make_heap(heap.begin(), heap.end(), worse);
        while (have < want) {
            std::pop_heap(heap.begin(), heap.end(), worse);
            stretch &x = st[heap.back()];
            if ((x.r1 - x.r0) / (x.k + inc) < (twin ? 8 : 12)) break;   // (units shorter than their own fill: the most expensive stretch cannot be helped further)
            x.k += inc; have += inc;
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    }
    for (const stretch &x : st) {
        const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
        int r = x.r0;
        double acc = 0.0;
        if (twin) continue;                                   // (below)
        for (int u = 0; u < x.k; u++) {
            const int u0 = r;
            const double upto = x.cost * (u + 1) / x.k;
            while (r < x.r1 && (u == x.k - 1 || acc + 0.5 * roww(cls, r) <= upto) && r - u0 < ST_MAXROWS) { acc += roww(cls, r); r++; }
            if (u == x.k - 1 && r < x.r1) {                       // (the row limit cut the last unit short: the rest as further units)
                emit(x.tx, cls, u0, r, flipping && (u & 1));
                for (int q = r; q < x.r1; q += ST_MAXROWS) emit(x.tx, cls, q, std::min(q + ST_MAXROWS, x.r1), 0);
                r = x.r1;
                break;
            }
            emit(x.tx, cls, u0, r, flipping && (u & 1));
        }
    }
    if (twin) {
        // Twin units.  A stretch's k units end at rows c[1] .. c[k]; units 2j (southwards) and 2j + 1 (northwards) meet on the
        // seam c[2j + 1].  What a unit costs is not a sum over its rows -- one coast row puts the five steps it is in flight for
        // through the coast formula, a land row makes them tested steps, a unit's steps come in pairs -- so units are priced as
        // the kernel will run them (st_exact_cost), the stretches are packed under a common limit (below), and the boundaries are
        // then moved, up to four rows at a time, towards the dearer of the two units they separate.
        struct cutline { std::vector<int> c; float worst = 0.0f; };
        auto unit_cost = [&](const stretch &x, const std::vector<int> &c, int u) -> float {
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            int a = c[u], b = c[u + 1];
            if (u & 1) { while (cls && b > a && cls[b - 1] == 1) b--; }      // (the far ends are trimmed: emit_twin)
            else { while (cls && a < b && cls[a] == 1) a++; }
            return st_exact_cost(o, x.tx, a, b, !(u & 1), true);
        };
        auto first_cut = [&](const stretch &x, std::vector<int> &c) {
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            c.assign((size_t)x.k + 1, x.r1);
            c[0] = x.r0;
            int r = x.r0;
            double acc = 0.0;
            bool capped = false;
            for (int u = 0; u < x.k; u++) {
                const int u0 = r;
                const double upto = x.cost * (u + 1) / x.k;
                while (r < x.r1 && (u == x.k - 1 || acc + 0.5 * roww(cls, r) <= upto)) { acc += roww(cls, r); r++; }
                c[u + 1] = r;
                capped |= r - u0 > ST_TWIN_MAXROWS;
            }
            if (capped)                                           // (a unit of mostly land grew past the row limit: equal rows instead)
                for (int u = 1; u < x.k; u++) c[u] = x.r0 + (int)((long long)(x.r1 - x.r0) * u / x.k);
        };
        auto refine = [&](const stretch &x, cutline &cl) {
            std::vector<int> &c = cl.c;
            std::vector<float> cost((size_t)x.k);
            for (int u = 0; u < x.k; u++) cost[u] = unit_cost(x, c, u);
            for (int sweep = 0; sweep < 48; sweep++) {
                bool moved = false;
                for (int b = 1; b < x.k; b++) {
                    // (a unit's steps come in pairs, four rows: its cost is a staircase in its rows, so moves of up to four rows are tried)
                    const int dir = cost[b - 1] > cost[b] ? -1 : 1;          // shrink the dearer unit
                    const int old = c[b];
                    float best_max = std::max(cost[b - 1], cost[b]) - 1e-4f, best_a = 0.0f, best_b = 0.0f;
                    int best_nb = old;
                    for (int step = 1; step <= 4; step++) {
                        const int nb = old + dir * step;
                        if (nb - c[b - 1] < 1 || c[b + 1] - nb < 1 || nb - c[b - 1] > ST_TWIN_MAXROWS || c[b + 1] - nb > ST_TWIN_MAXROWS) break;
                        c[b] = nb;
                        const float ca = unit_cost(x, c, b - 1), cb = unit_cost(x, c, b);
                        if (std::max(ca, cb) < best_max) { best_max = std::max(ca, cb); best_a = ca; best_b = cb; best_nb = nb; }
                    }
                    c[b] = best_nb;
                    if (best_nb != old) { cost[b - 1] = best_a; cost[b] = best_b; moved = true; }
                }
                if (!moved) break;
            }
            cl.worst = 0.0f;
            for (int u = 0; u < x.k; u++) cl.worst = std::max(cl.worst, cost[u]);
        };
        std::vector<cutline> cuts(st.size());
        // How many units a stretch gets: the smallest limit T is looked for (bisection) under which the stretches, each packed
        // greedily into units that cost at most T -- a unit grows by four rows, one iteration of the kernel's loop, while it stays
        // under T -- need no more than `want` units together.  (Dealt out by the stretches' summed row weights, the stretches along
        // the coasts came away with too few: their units end on land, and a unit that ends on land has five tested steps
        // whatever its height -- forcing the coast weight to 2.4 in that deal brought the 8-way band from 52.9 to 48.1 us.)
        static const int pack_env = [] { const char *e = dmg_hook("DMG_ST_PACK"); return e ? atoi(e) : 1; }();
        auto cost_rows = [&](const stretch &x, int a, int b, int u) -> float {
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            if (u & 1) { while (cls && b > a && cls[b - 1] == 1) b--; }
            else { while (cls && a < b && cls[a] == 1) a++; }
            return st_exact_cost(o, x.tx, a, b, !(u & 1), true);
        };
        auto pack = [&](const stretch &x, float T, std::vector<int> *c) -> int {
            int r = x.r0, u = 0;
            if (c) { c->clear(); c->push_back(r); }
            while (r < x.r1) {
                int n = std::min(4, x.r1 - r);
                while (r + n < x.r1 && n + 4 <= ST_TWIN_MAXROWS && cost_rows(x, r, std::min(r + n + 4, x.r1), u) <= T) n = std::min(n + 4, x.r1 - r);
                if (x.r1 - (r + n) < 3 && x.r1 - r <= ST_TWIN_MAXROWS) n = x.r1 - r;     // (no sliver at the end)
                r += n; u++;
                if (c) c->push_back(r);
            }
            if (u & 1) { u++; if (c) c->push_back(x.r1); }               // (pairs: an empty unit as the last one's partner)
            return u;
        };
        if (pack_env) {
            float t_lo = 8.0f, t_hi = 8.0f;
            for (const stretch &x : st) t_hi = std::max(t_hi, cost_rows(x, x.r0, std::min(x.r1, x.r0 + ST_TWIN_MAXROWS), 1));
            std::vector<int> packed(st.size());
            auto total = [&](float T) {
                st_parallel_for(st.size(), [&](size_t i) { packed[i] = pack(st[i], T, nullptr); });
                long n = 0;
                for (int k : packed) n += k;
                return n;
            };
            if (total(t_hi) <= want)
                for (int it = 0; it < 12; it++) {
                    const float mid = 0.5f * (t_lo + t_hi);
                    if (total(mid) <= want) t_hi = mid; else t_lo = mid;
                }
            long have_units = 0;
            st_parallel_for(st.size(), [&](size_t i) { st[i].k = pack(st[i], t_hi, &cuts[i].c); refine(st[i], cuts[i]); });
            for (const stretch &x : st) have_units += x.k;
            // (units grow four rows at a time and come in pairs: the limit that fits leaves some of the round unused -- those go, a
            // pair at a time, to the stretch whose dearest unit is the dearest, while that makes it cheaper)
            std::vector<char> done(st.size(), 0);
            while (have_units + 2 <= want) {
                size_t w = st.size();
                for (size_t i = 0; i < st.size(); i++) if (!done[i] && (w == st.size() || cuts[i].worst > cuts[w].worst)) w = i;
                if (w == st.size()) break;
                stretch x = st[w];
                if ((x.r1 - x.r0) / (x.k + 2) < 4) { done[w] = 1; continue; }
                x.k += 2;
                cutline c2;
                first_cut(x, c2.c); refine(x, c2);
                if (c2.worst < cuts[w].worst - 1e-4f) { st[w] = x; cuts[w] = c2; have_units += 2; }
                else done[w] = 1;
            }
        }
        else st_parallel_for(st.size(), [&](size_t i) { first_cut(st[i], cuts[i].c); refine(st[i], cuts[i]); });
        for (size_t i = 0; i < st.size(); i++) {
            const stretch &x = st[i];
            const unsigned char *cls = classes ? table.data() + (size_t)x.tx * L : nullptr;
            for (int j = 0; j + 1 < x.k; j += 2) emit_twin(x.tx, cls, cuts[i].c[j], cuts[i].c[j + 1], cuts[i].c[j + 2]);
        }
    }
}

// makespan of the units on `slots` waves that each take the next unit when they are done (how the dispatcher hands
// workgroups to free slots), units in descending cost order.
//
// Wide form (two waves per SIMD): the two waves of a SIMD do not share it evenly.  Per-unit timings with the hardware ids
// (DMG_ST_PROF) show the dispatcher filling wave slot 0 of every SIMD with the launch's first 4 x CUs workgroups and slot 1
// with the next, and the SIMD's arbiter serving the older wave first: of two equal units started together the first takes
// 195 us and the second 248 us.  The model: the older (or only) wave of a SIMD advances at 1.25, the younger at 0.75 of the
// even share; when a wave ends, the other one is the older from then on and the next unit of the list starts as the
// younger.  Against a measured 455 / 361 us (no land / survey terrain) it predicts 435 / 357 us, where one speed for all
// slots was 10 % optimistic about the tail and preferred cuts accordingly.
static float st_makespan(std::vector<st_unit> &units, int slots, bool paired = false)
{
    std::sort(units.begin(), units.end(), [](const st_unit &a, const st_unit &b) { return a.cost > b.cost; });
    if (!paired || slots < 2) {
        std::vector<float> heap((size_t)slots, 0.0f);      // min-heap of slot finish times
        auto cmp = [](float a, float b) { return a > b; };
        float span = 0.0f;
        for (const st_unit &u : units) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            heap.back() += u.cost;
            span = std::max(span, heap.back());
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        return span;
    }
    static const float r_old_env = [] { const char *e = dmg_hook("DMG_ST_ROLD"); return e ? (float)atof(e) : 1.25f; }();     // experiments (1.15 .. 1.35 choose the same or equally fast cuts)
    const float r_old = r_old_env, r_young = 2.0f - r_old_env;
    const int nsimd = slots / 2, n = (int)units.size();
    struct simd { double rem[2]; double t; signed char busy[2]; signed char older; };
    std::vector<simd> sm((size_t)nsimd, simd{ { 0.0, 0.0 }, 0.0, { 0, 0 }, 0 });
    typedef std::pair<double, int> ev;                      // (time of the SIMD's next completion, SIMD)
    std::vector<ev> heap;
    heap.reserve((size_t)nsimd + 16);
    auto later = [](const ev &a, const ev &b) { return a.first > b.first; };
    auto schedule = [&](int s) {
        simd &m = sm[s];
        double dt;
        if (m.busy[0] && m.busy[1]) dt = std::min(m.rem[m.older] / r_old, m.rem[1 - m.older] / r_young);
        else if (m.busy[0] || m.busy[1]) dt = m.rem[m.busy[0] ? 0 : 1] / r_old;
        else return;
        heap.push_back(ev(m.t + dt, s));
        std::push_heap(heap.begin(), heap.end(), later);
    };
    int next = 0;
    for (int k = 0; k < 2; k++)
        for (int s = 0; s < nsimd && next < n; s++) { sm[s].rem[k] = units[next++].cost; sm[s].busy[k] = 1; }
    for (int s = 0; s < nsimd; s++) schedule(s);
    float span = 0.0f;
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), later);
        const ev e = heap.back();
        heap.pop_back();
        simd &m = sm[e.second];
        const double dt = e.first - m.t;
        if (m.busy[0] && m.busy[1]) { m.rem[m.older] -= dt * r_old; m.rem[1 - m.older] -= dt * r_young; }
        else if (m.busy[0] || m.busy[1]) m.rem[m.busy[0] ? 0 : 1] -= dt * r_old;
        m.t = e.first;
        for (int k = 0; k < 2; k++)
            if (m.busy[k] && m.rem[k] <= 1e-6) {
                m.busy[k] = 0;
                span = std::max(span, (float)m.t);
                if (m.busy[1 - k]) m.older = (signed char)(1 - k);
                if (next < n) {
                    m.rem[k] = units[next++].cost; m.busy[k] = 1;
                    if (!m.busy[1 - k]) m.older = (signed char)k;
                }
            }
        schedule(e.second);
    }
    return span;
}

// Where the workgroups of a one-round launch of twin units run, and next to whom.  Per-unit timings with the hardware ids
// (DMG_ST_PROF, profiles/r04_twin_units.txt; 256 of 256 CUs alike): workgroup i of a launch of 128-thread workgroups runs on
// XCD (i + const) mod 8 and on the CU that also gets workgroups i + 256, i + 512, i + 768 (four fit: LDS), and the CU deals their
// eight waves w0 .. w7 (two per workgroup, in order) to its SIMDs as  S, S+1 | S+1, S+2 | S+2, S+3 | S+3, S  -- so w1 and w2
// share a SIMD, w3 and w4, w5 and w6, w7 and w0, the first of each couple being the older wave there, the one the SIMD's
// arbiter serves first.  A SIMD is done when both its waves are: a launch of one round lasts as long as the SIMD with the
// most work, and with the workgroups in longest-first order that is the one whose older wave came from the dearest quarter
// of the list and its younger wave from the next (predicted sums 34.6 .. 45.0 around a median of 38.0, and the twelve units
// that ended last all sat on such SIMDs).  Here the list is ORDERED for that hardware: every XCD gets the pairs of a contiguous
// range of strips (neighbouring strips share halo columns in its L2, as before); within it the dearest pairs become the CUs'
// first workgroups, the cheapest their fourth, matched dearest with cheapest, the second and third likewise; and within a CU the
// order of the two units inside each workgroup (which of them is the first wave) is chosen, all 16 ways tried, so that the
// four SIMDs' sums are as equal as they can be, the older wave the dearer one where that costs nothing.  If the dispatcher
// ever deals differently the order is merely arbitrary -- no worse than any other.  Lists of fewer than four workgroups per CU
// are padded with workgroups that end at once, so that position i + 256 k stays on CU i.
static void st_place_twins(std::vector<st_unit> &units, int num_cu, int tiles_x)
{
    const int C = std::max(num_cu, 8) / 8 * 8, per_x = C / 8;          // CUs, CUs per XCD
    const int n = (int)units.size();
    const int R = (n + C - 1) / C;                                      // workgroups per CU
    if (n == 0 || R > 4) return;
    std::vector<st_unit> sorted = units;
    std::sort(sorted.begin(), sorted.end(), [](const st_unit &a, const st_unit &b) { return a.tx != b.tx ? a.tx < b.tx : a.r0 < b.r0; });
    st_unit dummy{ 0, 0, 0, 0.0f, 0 };
    dummy.rlo = 0; dummy.dummy = 1;
    std::vector<st_unit> out((size_t)R * C, dummy);
    double couple_lo = 1e30, couple_hi = 0.0;
    for (int x = 0; x < 8; x++) {
        // this XCD's pairs: a contiguous eighth of the list in strip order
        const int g0 = (int)((long long)n * x / 8), g1 = (int)((long long)n * (x + 1) / 8);
        std::vector<st_unit> g(sorted.begin() + g0, sorted.begin() + g1);
        const int m = (int)g.size();
        if (!m) continue;
        // Every workgroup's dearer unit is its SECOND wave: the couples of a CU are then (dearer unit of workgroup r, cheaper unit of
        // workgroup r + 1), cyclically, the dearer one the older wave of its SIMD.  The CU's workgroups are chosen as a cycle whose
        // couples all come close to the mean sum: start from the unused pair with the dearest unit, follow it with the pair whose
        // cheaper unit complements that best, and so on; the last one also has to suit the first.
        auto hi = [](const st_unit &u) { return std::max(u.cost_n, u.cost_s); };
        auto lo = [](const st_unit &u) { return std::min(u.cost_n, u.cost_s); };
        double target = 0.0;
        for (const st_unit &u : g) target += u.cost_n + u.cost_s;
        target /= m;                                                    // the mean couple
        std::vector<char> used((size_t)m, 0);
        int left = m;
        for (int t = 0; t < per_x && left > 0; t++) {
            const int len = std::min(left, (left + (per_x - t) - 1) / (per_x - t));      // this CU's share of what is left (at most R)
            int cyc[4] = { -1, -1, -1, -1 };
            for (int r = 0; r < len; r++) {
                int pick = -1;
                double best_d = 1e30;
                for (int i = 0; i < m; i++) {
                    if (used[i]) continue;
                    double dd;
                    if (r == 0) dd = -hi(g[i]);                        // the dearest unit left opens the cycle
                    else {
                        dd = std::fabs(hi(g[cyc[r - 1]]) + lo(g[i]) - target);
                        if (r == len - 1) dd = std::max(dd, std::fabs(hi(g[i]) + lo(g[cyc[0]]) - target));
                    }
                    if (dd < best_d) { best_d = dd; pick = i; }
                }
                cyc[r] = pick; used[pick] = 1; left--;
            }
            for (int r = 0; r < len; r++) {
                st_unit u = g[cyc[r]];
                u.south_first = u.cost_s < u.cost_n ? 1 : 0;           // the cheaper unit first
                out[(size_t)r * C + 8 * t + x] = u;
                const double couple = hi(g[cyc[r]]) + lo(g[cyc[(r + 1) % len]]);
                couple_lo = std::min(couple_lo, couple); couple_hi = std::max(couple_hi, couple);
            }
        }
    }
    // (trailing dummies are dropped: nothing follows them)
    while (!out.empty() && out.back().dummy) out.pop_back();
    static const bool dbg = dmg_hook("DMG_ST_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[stream]   %d pairs placed on %d CUs, %d per CU: the SIMDs' predicted sums %.1f .. %.1f steps\n", n, C, R, couple_lo, couple_hi);
    units.swap(out);
}

// The unit list of a launch geometry.  A launch has only 2 x 4 x CUs wave slots and its units all take about the same
// time, so the chip works in "rounds": 2.49 rounds of 96-row units cost as much as 3.  The unit height is therefore chosen
// per geometry (and terrain) by simulating the dispatch for every candidate height -- taller units recompute less
// (7/n), but the last round must be full -- and the units are issued longest first.
// part 1 / 2: only the units that do / do not produce rows of [lo, zone_lo) or [zone_hi, hi) (the two halves of a launch
// split around a halo exchange).
static thread_local bool st_units_failed_flag = false;         // the last st_units call of this thread failed with a message of its own
bool st_units_failed() { return st_units_failed_flag; }
const dmg_ocean::tile_list *st_units(dmg_ocean *o, int variant, int lo, int hi, int part, int zone_lo, int zone_hi, int seg)
{
    st_units_failed_flag = false;
    auto key = std::make_tuple(lo, hi, -1000 - seg, part, zone_lo, zone_hi);    // "rows per wave" <= -1000 = stream units (seg 0 = auto)
    auto it = o->tb_lists.find(key);
    if (it != o->tb_lists.end()) return &it->second;
    const int slots = std::max(o->ctx->num_cu, 1) * 8;     // two waves per SIMD
    std::vector<st_unit> units, best;
    int cut_seg = 0, cut_seg2 = 0, cut_split = 0;         // the cut this list was made with (st_cut's seg, seg2, split)
    int cut_twin = 0;                                     // ... a balanced cut into twin units
    if (part) {
        // The two halves of a split launch are halves of ONE cut: the one the whole launch over these rows uses (chosen
        // with every unit in play), filtered by the edge predicate.  (Each half choosing its own unit heights left the
        // rows between the two choices to neither of them.)
        const dmg_ocean::tile_list *whole = st_units(o, variant, lo, hi, 0, 0, 0, seg);
        if (!whole) return nullptr;
        cut_seg = whole->cut_seg; cut_seg2 = whole->cut_seg2; cut_split = whole->cut_split; cut_twin = whole->cut_twin;
        if (cut_seg == 0) { cut_seg = ST_MAXROWS; cut_seg2 = 0; cut_split = hi; }   // (a launch with no live unit)
        static const int edge_seg = [] { const char *e = dmg_hook("DMG_ST_EDGE_SEG"); const int v = e ? atoi(e) : 4; return v >= 2 && v <= ST_MAXROWS ? v : 0; }();
        auto cut_as_whole = [&](int which, std::vector<st_unit> &out) {      // (cut_seg < 0: the balanced cut, -cut_seg units in all)
            if (cut_seg < 0) st_cut_balanced(o, lo, hi, -cut_seg, which, zone_lo, zone_hi, out, edge_seg, cut_twin != 0);
            else st_cut(o, variant, lo, hi, cut_seg, which, zone_lo, zone_hi, out, cut_seg2, cut_split, edge_seg);
        };
        cut_as_whole(part, best);
        (void)st_makespan(best, slots);
        auto units_of = [](const std::vector<st_unit> &v) { size_t n = 0; for (const st_unit &u : v) n += u.dummy ? 0 : (u.rlo >= 0 ? 2 : 1); return n; };
        // both halves built: together they must be the units of ONE cut of the launch's rows, each exactly once
        auto other = o->tb_lists.find(std::make_tuple(lo, hi, std::get<2>(key), 3 - part, zone_lo, zone_hi));
        if (other != o->tb_lists.end()) {
            std::vector<st_unit> all;
            cut_as_whole(0, all);
            long long rows_all = 0, rows_parts = 0;
            for (const st_unit &u : all) rows_all += u.r1 - u.first();
            for (const st_unit &u : best) rows_parts += u.r1 - u.first();
            if (other->second.n_cut_units + units_of(best) != units_of(all) || other->second.cut_rows + rows_parts != rows_all) {
                st_units_failed_flag = true; dmg_fail(DMG_ESTATE, "split launch over rows [%d, %d): %zu + %zu units in the two parts, %zu in the cut they are halves of", lo, hi,
                         other->second.n_cut_units, units_of(best), units_of(all));
                return nullptr;
            }
        }
    }
    else if (seg > 0) { st_cut(o, variant, lo, hi, seg, part, zone_lo, zone_hi, best); (void)st_makespan(best, slots); cut_seg = seg; cut_split = hi; }
    else {
        float best_span = 1e30f;
        int best_seg = 0, best_seg2 = 0, best_split = 0;
        // what a cut is judged by: alpha * (makespan of the simulated dispatch) + (1 - alpha) * (its work / slots).  The chip's
        // throughput is shared by the two waves of a SIMD -- when a slot idles, its neighbour speeds up -- so an idle tail costs
        // less than its length and the work (taller units stream fewer fill rows) counts for more than the makespan alone says
        static const float alpha_env = [] { const char *e = dmg_hook("DMG_ST_ALPHA"); return e ? (float)atof(e) : -1.0f; }();
        const float alpha = alpha_env >= 0.0f ? alpha_env : (seg == -2 ? 0.5f : 1.0f);                  // seg -2: the second automatic cut
        auto judge = [&](std::vector<st_unit> &u) {
            const float span = st_makespan(u, slots, variant == 0);
            double total = 0;
            for (const st_unit &x : u) total += x.cost;
            return alpha * span + (1.0f - alpha) * (float)(total / slots);
        };
        // (every height up to 48 rows: a band of a thousand rows has ~25 rows per wave slot, and one row more or less per unit
        // decides whether the launch's single round fills the slots or spills into a second)
        static const int fine_env = [] { const char *e = dmg_hook("DMG_ST_FINE"); return e ? atoi(e) : 1; }();
        // The candidates -- every height, and for heights of 48 rows and more two tiers: the top eighth .. half of the rows in units
        // a half, a third or a quarter as tall -- are judged side by side on the host's cores (640 cuts and simulated dispatches:
        // 1.2 s on one core of the build container for the full grid; VERDICT r4: an editor that changes the terrain pays this per
        // launch geometry).  The winner is the first of the smallest in the order a single thread would have tried them.
        const auto t_plan0 = std::chrono::steady_clock::now();
        struct st_cand { int seg, seg2, split; float span; };
        std::vector<st_cand> cands;
        for (int cand = fine_env ? 20 : 24; cand <= ST_MAXROWS; cand += (fine_env && cand < 48) ? 1 : 4) {
            cands.push_back(st_cand{ cand, 0, hi, 0.0f });
            for (int frac = 1; frac <= 4 && cand >= 48; frac++)
                for (int div = 2; div <= 4; div++) {
                    const int split = hi - ((hi - lo) * frac / 8 / cand) * cand, seg2 = ((cand / div + 3) / 4) * 4;
                    if (split <= lo || split >= hi || seg2 < 12) continue;
                    cands.push_back(st_cand{ cand, seg2, split, 0.0f });
                }
        }
        st_parallel_for(cands.size(), [&](size_t i) {
            std::vector<st_unit> mine;
            st_cand &c = cands[i];
            st_cut(o, variant, lo, hi, c.seg, part, zone_lo, zone_hi, mine, c.seg2, c.seg2 ? c.split : 0);
            c.span = judge(mine);
        });
        for (const st_cand &c : cands)
            if (c.span < best_span) { best_span = c.span; best_seg = c.seg; best_seg2 = c.seg2; best_split = c.split; }
        if (best_seg) {
            st_cut(o, variant, lo, hi, best_seg, part, zone_lo, zone_hi, best, best_seg2, best_seg2 ? best_split : 0);
            (void)judge(best);
        }
        if (dmg_hook("DMG_ST_PLAN_TIMING"))               // experiments
            fprintf(stderr, "[stream] rows [%d,%d): %zu candidate cuts judged in %.1f ms\n", lo, hi, cands.size(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count());
        // ... and the strip-by-strip cut into units of equal cost (st_cut_balanced), at one, two and three rounds' worth of units
        static const int bal_env = [] { const char *e = dmg_hook("DMG_ST_BALANCED"); return e ? atoi(e) : 1; }();     // how many rounds' worth at most (0: off)
        // (only on launches of at most ST_BALANCED_ROWS rows whose best fixed-height cut is itself about one round -- the band of an
        // 8-way split.  There it is worth 5-9 % (16384 x 1024: 65.0 -> 59.5 / 61.6 us on two boxes; the middle band of the 8-way split
        // 62.3 -> 59.3 us).  On taller launches the simulation's preference is not the chip's: 16384 x 2048 in one round of 2048
        // equal-cost units takes 94 us against 87 for 1981 units of 52 rows -- strips whose units no longer cover the same rows at
        // the same time stop sharing their halo columns in the L2 -- although the 4-way middle band gains 4.5 %;
        // profiles/r04_balanced_cut.txt)
        static const int bal_rows = [] { const char *e = dmg_hook("DMG_ST_BALANCED_ROWS"); return e ? atoi(e) : 1200; }();
        if (bal_env && variant == 0 && !o->rowclass_host.empty() && (int)best.size() <= slots + slots / 10 && hi - lo <= bal_rows)
            for (int eighths = 6; eighths <= 8 * bal_env; eighths += (eighths < 8 ? 1 : 8)) {      // 3/4, 7/8, 1 (, 2, 3) rounds' worth of units
                const int want = slots * eighths / 8;
                st_cut_balanced(o, lo, hi, want, part, zone_lo, zone_hi, units);
                if (units.empty()) continue;
                const float span = judge(units);
                static const bool dbgc = dmg_hook("DMG_ST_DEBUG") != nullptr;
                if (dbgc) fprintf(stderr, "[stream]   balanced cut of %d units: predicted span %.0f (best so far %.0f)\n", want, span, best_span);
                if (span < best_span) { best_span = span; best.swap(units); best_seg = -want; best_seg2 = 0; best_split = hi; }
            }
        // ... and into twin units (pairs that start on a common seam and need no halo between them: k_jacobi_stream_twin), one round's worth
        // (read whenever a list is built, not once per process: bench.py switches the twin units off and tries again if every leg
        // of a first multi-GPU run fails with them)
        const int twin_env = [] { const char *e = getenv("DMG_ST_TWIN"); return e ? atoi(e) : ST_TWIN_DEFAULT; }();
        // (up to the band of a 2-way split.  16384 x 2048: one round of twin units 73.1 us against 81.6 on the survey terrain, 93.9 against
        // 114.2 without land; 16384 x 4096: 144.2-145.6 against 149.9 / 187.3 against 188.5.  At 16384 x 8192 they LOSE to 1.3 rounds of
        // 200-row plain units, as two or three rounds of short pairs (292-296 against 273.9 us, 395 against 362.6 without land) and as
        // one round of pairs of up to 232 rows (300.2 against 279.9 us): a launch of one round lasts as long as its slowest SIMD, 1.2
        // medians, where the plain cut's short units fill the slots its tall ones leave -- profiles/r04_twin_units.txt)
        static const int twin_rows = [] { const char *e = dmg_hook("DMG_ST_TWIN_ROWS"); return e ? atoi(e) : 4400; }();
        if (twin_env && bal_env && variant == 0 && !o->rowclass_host.empty() && hi - lo <= twin_rows) {
            // (the twin units' costs are st_exact_cost's: the cut they are compared with is priced the same way)
            std::vector<st_unit> ref = best;
            for (st_unit &u : ref) u.cost = 16.0f * st_exact_cost(o, u.tx, u.r0, u.r1, u.flip != 0, false);
            const float ref_span = ref.empty() ? 1e30f : st_makespan(ref, slots, true);
            float twin_span = 1e30f;
            std::vector<st_unit> twin_best;
            int twin_want = 0;
            // one round's worth of units (DMG_ST_TWIN_ROUNDS=3: also one and a half, two and three rounds' worth -- measured, never better)
            static const int twin_rounds = [] { const char *e = dmg_hook("DMG_ST_TWIN_ROUNDS"); return e ? atoi(e) : 1; }();
            for (int eighths = 8; eighths <= 8 * twin_rounds; eighths += (eighths < 16 ? 4 : 8)) {
                const int want = slots * eighths / 8;
                const auto t_cut0 = std::chrono::steady_clock::now();
                st_cut_balanced(o, lo, hi, want, part, zone_lo, zone_hi, units, 0, true);
                const double cut_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cut0).count();
                if (units.empty()) continue;
                const float span = st_makespan(units, slots / 2);          // (a pair holds two of the slots)
                static const int dbgc = [] { const char *e = dmg_hook("DMG_ST_DEBUG"); return e ? atoi(e) : 0; }();
                if (dbgc) {
                    double sum = 0; float lo_c = 1e30f;
                    for (const st_unit &u : units) { sum += u.cost; lo_c = std::min(lo_c, u.cost); }
                    fprintf(stderr, "[stream]   balanced cut into %zu pairs of twin units (%d asked for, %.0f ms on the host): predicted span %.0f (dearest pair %.0f, mean %.0f, cheapest %.0f); the best plain cut priced alike: %.0f\n",
                            units.size(), want / 2, cut_ms, span, units.front().cost, sum / units.size(), lo_c, ref_span);
                    for (size_t i = 0; i < units.size() && i < 12 && dbgc > 1; i++) {
                        const st_unit &u = units[i];
                        const unsigned char *cls = o->rowclass_host.data() + (size_t)u.tx * o->L;
                        const st_features fn = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, u.r0, u.r1, false, true);
                        const st_features fs = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, o->L - u.r0, o->L - u.rlo, true, true);
                        fprintf(stderr, "[stream]     pair %zu: strip %d rows [%d, %d | %d) cost %.0f; north: %d fast %d coast %d tested (%d/%d/%d) fill (%d/%d/%d) = %.1f; south: %d fast %d coast %d tested (%d/%d/%d) fill (%d/%d/%d) = %.1f\n",
                                i, u.tx, u.rlo, u.r0, u.r1, u.cost, fn.fast, fn.coast, fn.tested, fn.t_open, fn.t_mixed, fn.t_land, fn.f_open, fn.f_mixed, fn.f_land,
                                st_exact_cost(o, u.tx, u.r0, u.r1, false, true), fs.fast, fs.coast, fs.tested, fs.t_open, fs.t_mixed, fs.t_land, fs.f_open, fs.f_mixed, fs.f_land,
                                st_exact_cost(o, u.tx, u.rlo, u.r0, true, true));
                    }
                }
                if (span < twin_span) { twin_span = span; twin_best.swap(units); twin_want = want; }
            }
            if (!twin_best.empty() && (twin_span < ref_span || twin_env > 1)) { best_span = twin_span; best.swap(twin_best); best_seg = -twin_want; best_seg2 = 0; best_split = hi; cut_twin = 1; }
        }
        (void)st_makespan(best, cut_twin ? slots / 2 : slots);      // (longest first)
        cut_seg = best_seg; cut_seg2 = best_seg2; cut_split = best_split;
        static const bool dbg = dmg_hook("DMG_ST_DEBUG") != nullptr;
        if (dbg) {
            double total = 0;
            for (const st_unit &u : best) total += u.cost;
            fprintf(stderr, "[stream] rows [%d,%d) part %d: %zu units, %s %d rows up to row %d then %d; predicted span %.0f, mean load %.0f (%.1f %% of the slots' time)\n",
                    lo, hi, part, best.size(), best_seg < 0 ? "balanced cut of" : "", best_seg < 0 ? -best_seg : best_seg, best_split, best_seg2, best_span, total / slots,
                    100.0 * total / slots / best_span);
            // which path the steps of these units take (host classes: 0 open, 1 land, 2 mixed), and how many of the rows that go
            // through the coast formula in an all-coast step are open water
            const std::vector<unsigned char> &table = o->rowclass_host;
            if (!table.empty() && !cut_twin) {
                long long n1 = 0, n2 = 0, n0 = 0, rows2 = 0, open2 = 0;
                for (const st_unit &u : best) {
                    const unsigned char *cls = table.data() + (size_t)u.tx * o->L;
                    const int e0 = u.r0 - TB_T, nst = (((u.r1 - u.r0) + 2 * TB_T + 3) / 4) * 2;
                    for (int t = TB_T; t < nst; t++) {
                        bool land = false, mix = false; int nopen = 0;
                        for (int k = 0; k < 10; k++) {
                            const int r = e0 + 2 * t + 1 - k;
                            const int c = (r >= 0 && r < o->L) ? cls[r] : 2;
                            land |= c == 1; mix |= c == 2; nopen += c == 0;
                        }
                        if (land) n0++; else if (mix) { n2++; rows2 += 10; open2 += nopen; } else n1++;
                    }
                }
                fprintf(stderr, "[stream]   steps past the fill: %.1f %% open (no test), %.1f %% all through the coast formula, %.1f %% tested; of the rows in flight in coast steps %.1f %% are open water\n",
                        100.0 * n1 / (n1 + n2 + n0 + 1e-9), 100.0 * n2 / (n1 + n2 + n0 + 1e-9), 100.0 * n0 / (n1 + n2 + n0 + 1e-9), 100.0 * open2 / (rows2 + 1e-9));
            }
        }
    }
    {
        // Workgroup i of a launch runs on XCD (i + const) mod 8 (per-unit hardware ids, DMG_ST_PROF), each XCD behind its own L2.
        // Among units of equal cost -- whose order longest-first leaves open -- every XCD gets a contiguous range of strips,
        // row block by row block: the 16 halo columns two neighbouring strips share are then read through one L2 instead of
        // two (+0.7 % on the survey terrain, +1.0 % without land in a same-box A/B; wider cost buckets disturb longest-first
        // and lose).  DMG_ST_XCC=0 keeps the sorted order.
        static const int xcc_mode = [] { const char *e = dmg_hook("DMG_ST_XCC"); return e ? atoi(e) : 1; }();
        static const int place_mode = [] { const char *e = dmg_hook("DMG_ST_PLACE"); return e ? atoi(e) : 1; }();
        const bool twins = !best.empty() && best.front().rlo >= 0;
        if (twins && place_mode && (int)best.size() <= slots / 2) st_place_twins(best, o->ctx->num_cu, (o->grid.W + TB_OW - 1) / TB_OW);
        else if (xcc_mode && variant == 0 && !best.empty()) {
            const int tiles_x = (o->grid.W + TB_OW - 1) / TB_OW;
            const float top = best.front().cost;
            std::vector<st_unit> out;
            out.reserve(best.size());
            size_t i0 = 0;
            while (i0 < best.size()) {
                size_t i1 = i0;
                const float floor_cost = best[i0].cost - 1e-4f * top;                     // a run of equal costs
                while (i1 < best.size() && best[i1].cost >= floor_cost) i1++;
                std::vector<st_unit> lists[8];
                for (size_t i = i0; i < i1; i++) lists[std::min(7, best[i].tx * 8 / std::max(tiles_x, 1))].push_back(best[i]);
                for (int x = 0; x < 8; x++)
                    std::sort(lists[x].begin(), lists[x].end(), [](const st_unit &a, const st_unit &b) { return a.r0 != b.r0 ? a.r0 < b.r0 : a.tx < b.tx; });
                size_t pos[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                for (size_t done = 0; done < i1 - i0;)
                    for (int x = 0; x < 8; x++) {
                        // (keep the slot of XCD x even when its list has run dry: take from the longest remaining one)
                        int y = x;
                        if (pos[y] >= lists[y].size()) { size_t bestn = 0; for (int q = 0; q < 8; q++) if (lists[q].size() - pos[q] > bestn) { bestn = lists[q].size() - pos[q]; y = q; } if (!bestn) break; }
                        out.push_back(lists[y][pos[y]++]);
                        done++;
                        if (done == i1 - i0) break;
                    }
                i0 = i1;
            }
            best.swap(out);
        }
    }
    std::vector<int> flat;
    flat.reserve(best.size() * 4);
    // a unit is an open-water unit (bit 0 of its fourth word; st_open_unit) if every row of its window, and every row its
    // steps request past the window, is plain open water with exact divisor pairs inside the grid and the stored band
    static const bool open_units = [] { const char *e = dmg_hook("DMG_ST_OPEN_UNITS"); return e && atoi(e) != 0; }();     // (measured: no gain; off by default)
    const std::vector<unsigned char> &ctab = o->rowclass_host;
    int n_open = 0, n_flip = 0, n_twin = 0;
    for (const st_unit &u : best) n_twin += u.rlo >= 0;
    if (n_twin && dmg_hook("DMG_ST_TWIN_FAIL")) {            // tests: a launch that would run twin units fails instead (bench.py's fall-back to plain units)
        st_units_failed_flag = true; dmg_fail(DMG_ESTATE, "twin units: injected failure (DMG_ST_TWIN_FAIL)");
        return nullptr;
    }
    if (n_twin && n_twin != (int)best.size()) { st_units_failed_flag = true; dmg_fail(DMG_ESTATE, "a unit list mixes twin units with plain ones"); return nullptr; }
    for (const st_unit &u : best) {
        int flag = 0;
        if (u.rlo >= 0) {                                  // the unit above the seam and the one below it in the mirrored view's rows
            if (u.r1 - u.r0 > ST_TWIN_MAXROWS || u.r0 - u.rlo > ST_TWIN_MAXROWS) { st_units_failed_flag = true; dmg_fail(DMG_ESTATE, "a twin unit of more than %d rows", ST_TWIN_MAXROWS); return nullptr; }
            // (bit 4: the units of a split launch's edge half -- the exchange waits for them: they keep a raised priority)
            const int urgent = part == 1 ? 16 : 0;
            const int north[4] = { u.tx, u.r0, u.r1, (u.dummy ? 4 | 8 : 4) | urgent }, south[4] = { u.tx, o->L - u.r0, o->L - u.rlo, (u.dummy ? 4 | 2 | 8 : 4 | 2) | urgent };
            flat.insert(flat.end(), u.south_first ? south : north, (u.south_first ? south : north) + 4);
            flat.insert(flat.end(), u.south_first ? north : south, (u.south_first ? north : south) + 4);
            n_flip++;
            continue;
        }
        if (open_units && variant == 0 && !ctab.empty()) {
            const int e0 = u.r0 - TB_T, nst = (((u.r1 - u.r0) + 2 * TB_T + 3) / 4) * 2, e_last = e0 + 2 * nst + 5;
            const int y_first = o->y0 - o->gs;
            bool ok = e0 >= 0 && e_last < o->L && y_first + e0 >= 0 && y_first + e_last < o->grid.H;
            const unsigned char *cls = ctab.data() + (size_t)u.tx * o->L;
            for (int r = e0; ok && r <= e0 + 2 * nst + 1; r++) ok = cls[r] == 0 && (o->rowbad_host.empty() || !o->rowbad_host[r]);
            flag = ok ? 1 : 0;
        }
        n_open += flag;
        if (u.flip && !flag) { flat.push_back(u.tx); flat.push_back(o->L - u.r1); flat.push_back(o->L - u.r0); flat.push_back(2); n_flip++; }   // the mirrored view's rows
        else { flat.push_back(u.tx); flat.push_back(u.r0); flat.push_back(u.r1); flat.push_back(flag); }
    }
    {
        static const bool dbg = dmg_hook("DMG_ST_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[stream] rows [%d,%d) part %d: %d of %zu units are open-water units, %d stream southwards\n", lo, hi, part, n_open, best.size(), n_flip);
    }
    dmg_ocean::tile_list tl;
    tl.n = (int)best.size();
    tl.twin = n_twin ? 1 : 0;
    tl.n_units = (size_t)(n_twin ? 2 * tl.n : tl.n);
    tl.n_cut_units = 0;
    for (const st_unit &u : best) tl.n_cut_units += u.dummy ? 0 : (u.rlo >= 0 ? 2 : 1);
    tl.n_flip = n_flip;
    tl.cut_seg = cut_seg; tl.cut_seg2 = cut_seg2; tl.cut_split = cut_split; tl.cut_twin = cut_twin;
    tl.cut_rows = 0;
    for (const st_unit &u : best) tl.cut_rows += u.r1 - u.first();
    if (o->lists_on_host) { tl.host = flat; return &(o->tb_lists[key] = tl); }       // (dmg_stream_units_probe: no device)
    if (tl.n && hipMalloc((void **)&tl.d, flat.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (tl.n && hipMemcpy(tl.d, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return &(o->tb_lists[key] = tl);
}


// Test hook (include/demiurge_hip.h): the unit list the streaming kernel would be launched with, from a row-class table, on
// the host alone -- the cut, the halves of a split launch and the placement are plain host code, and the multi-GPU paths that
// depend on them cannot run on a one-GPU box.
extern "C" int dmg_stream_units_probe(int W, int H, int y0, int y1, int gs, int gn, const unsigned char *rowclass, int num_cu,
                                      int lo, int hi, int part, int zone_lo, int zone_hi, int *out, size_t cap, size_t *n_entries, int *twin)
{
    if (W < TB_EW || H < 1 || y0 < 0 || y1 <= y0 || y1 > H || gs < 0 || gn < 0 || !rowclass || num_cu < 1 || !n_entries)
        return dmg_fail(DMG_EINVAL, "dmg_stream_units_probe: bad arguments");
    dmg_ctx ctx;
    ctx.num_cu = num_cu;
    dmg_ocean o;
    o.ctx = &ctx;
    o.grid.W = W; o.grid.H = H;
    o.y0 = y0; o.y1 = y1; o.gs = gs; o.gn = gn;
    o.L = gs + (y1 - y0) + gn;
    o.lists_on_host = true;
    if (lo < 0 || hi > o.L || lo >= hi) return dmg_fail(DMG_EINVAL, "dmg_stream_units_probe: rows [%d, %d) outside the band's %d", lo, hi, o.L);
    const int tiles_x = (W + TB_OW - 1) / TB_OW;
    o.rowclass_host.assign(rowclass, rowclass + (size_t)tiles_x * o.L);
    const dmg_ocean::tile_list *tl = st_units(&o, 0, lo, hi, part, zone_lo, zone_hi, 0);
    if (!tl) return st_units_failed() ? DMG_ESTATE : dmg_fail(DMG_ENOMEM, "unit list allocation failed");
    *n_entries = tl->host.size() / 4;
    if (twin) *twin = tl->twin;
    if (out && tl->host.size() <= cap * 4) memcpy(out, tl->host.data(), tl->host.size() * sizeof(int));
    else if (out) return dmg_fail(DMG_EINVAL, "dmg_stream_units_probe: %zu entries, room for %zu", tl->host.size() / 4, cap);
    return DMG_OK;
}
