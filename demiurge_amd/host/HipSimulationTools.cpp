#include "HipSimulationTools.h"
#include "HipAbi.h"

static dmg_grid grid_of(const Project *p)
{
    dmg_grid g{};
    g.W = p->getWidth(); g.H = p->getHeight();
    for (int i = 0; i < 4; i++) g.coords[i] = p->coords[i];
    g.circumference = p->circumference;
    return g;
}

// cpufilter::run (reference src/filter/cpufilter.cpp:41-222): uplift and scaled heights once, then 3N rounds of
// { FlowFilter ; erosion/uplift pass }.  The reference crosses to the GL thread several times per round and reads the
// heights back after every pass; here the rounds run on the worker thread with the heights resident on the device and
// the terrain layer is refreshed once per round (what the user sees, :104-108,:214).
void HipErosion::run()
{
    if (!hip_abi_matches(&error_)) { setProgress({ true, 1.0f }); return; }
    std::vector<float> h;
    dmg_grid g{};
    dispatchGPU([&](Project *p) { h = p->terrain; g = grid_of(p); });                    // :47-91 heightdata / updrift set-up
    dmg_ctx *ctx = nullptr;
    dmg_erosion *e = nullptr;
    auto fail = [&](const char *where) {
        error_ = std::string(where) + ": " + dmg_last_error();
        if (e) dmg_erosion_destroy(e);
        if (ctx) dmg_destroy(ctx);
        setProgress({ true, 1.0f });
    };
    if (!layer_matches(h, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; setProgress({ true, 1.0f }); return; }
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) return fail("dmg_create");
    if (dmg_erosion_create(ctx, &g, h.data(), N_, exponent_, slope_exponent_, factor_, dolakes_ > 0 ? 1 : 0, &e)) return fail("dmg_erosion_create");
    for (int i = 0; i < rounds_ && !cancelled(); i++) {                                  // :93
        if (dmg_erosion_rounds(e, 1)) return fail("dmg_erosion_rounds");
        if (dmg_erosion_get(e, h.data())) return fail("dmg_erosion_get");
        dispatchGPU([&](Project *p) { p->terrain = h; });                                // :213 heightdata = terrain->downloadDataRAW()
        setProgress({ false, (float)(i + 1) / (float)N_ });                              // :103 (sic: i/N, reaches 3.0)
    }
    dmg_erosion_destroy(e);
    dmg_destroy(ctx);
    setProgress({ true, 1.0f });                                                         // :221
}

// Temperature::run (reference src/filter/Temperature.cpp:47-162): every frame ten steps of the energy-balance model
// on the field kept in the terrain layer, heights in scratch1
void HipTemperature::run()
{
    if (!hip_abi_matches(&error_)) { setProgress({ true, 1.0f }); return; }
    std::vector<float> T, heights;
    dmg_grid g{};
    dispatchGPU([&](Project *p) { T = p->terrain; heights = p->scratch1; g = grid_of(p); });
    dmg_ctx *ctx = nullptr;
    dmg_climate *k = nullptr;
    auto fail = [&](const char *where) {
        error_ = std::string(where) + ": " + dmg_last_error();
        if (k) dmg_climate_destroy(k);
        if (ctx) dmg_destroy(ctx);
        setProgress({ true, 1.0f });
    };
    if (!layer_matches(T, g.W, g.H) || heights.size() != T.size()) { error_ = "terrain / scratch1 layers do not hold width*height texels"; setProgress({ true, 1.0f }); return; }
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) return fail("dmg_create");
    if (dmg_climate_create(ctx, &g, T.data(), heights.data(), &k)) return fail("dmg_climate_create");
    if (dmg_climate_set_params(k, &params_)) return fail("dmg_climate_set_params");
    int i = 0;
    for (int f = 0; f < frames_ && !cancelled(); f++) {                                  // :48 (the index advances 10 per frame plus the loop's ++, :151 and :48)
        if (dmg_climate_steps(k, i, per_frame_)) return fail("dmg_climate_steps");
        i += per_frame_;
        if (dmg_climate_get(k, T.data())) return fail("dmg_climate_get");
        dispatchGPU([&](Project *p) { p->terrain = T; });                                // the frame's result is what the canvas shows
        i++;                                                                             // `for (int i=0;...;i++)` around the closure
        setProgress({ false, (float)(f + 1) / (float)frames_ });
    }
    dmg_climate_destroy(k);
    dmg_destroy(ctx);
    setProgress({ true, 1.0f });
}

// FlowFilter::run (reference src/filter/FlowFilter.cpp:719-786): terrain := pow(upstream area, exponent)
void HipFlow::run()
{
    if (!hip_abi_matches(&error_)) { setProgress({ true, 1.0f }); return; }
    std::vector<float> h;
    dmg_grid g{};
    dispatchGPU([&](Project *p) { h = p->terrain; g = grid_of(p); });
    if (!layer_matches(h, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; setProgress({ true, 1.0f }); return; }
    dmg_ctx *ctx = nullptr;
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) { error_ = std::string("dmg_create: ") + dmg_last_error(); setProgress({ true, 1.0f }); return; }
    std::vector<float> flow(h.size());
    if (dmg_flow_accumulate(ctx, &g, h.data(), nullptr, preblur_, exponent_, lakeflag_ ? 1 : 0, flow.data()))
        error_ = std::string("dmg_flow_accumulate: ") + dmg_last_error();
    else
        dispatchGPU([&](Project *p) { p->terrain = flow; });                             // :714-716 uploadData into the terrain texture
    dmg_destroy(ctx);
    setProgress({ true, 1.0f });
}

std::pair<bool, float> HipOneShot::step(Project *p)
{
    dmg_grid g = grid_of(p);
    if (!layer_matches(p->terrain, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; return { true, 1.0f }; }
    dmg_ctx *ctx = nullptr;
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) { error_ = std::string("dmg_create: ") + dmg_last_error(); return { true, 1.0f }; }
    if (apply(ctx, &g, p->terrain.data())) error_ = dmg_last_error();
    dmg_destroy(ctx);
    return { true, 1.0f };
}
