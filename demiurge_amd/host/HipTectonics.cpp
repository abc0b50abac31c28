#include "HipTectonics.h"
#include "HipAbi.h"

// Tectonics::run (reference src/filter/tectonics/Tectonics.cpp:156-275): the reference dispatches one closure per
// step onto the GL thread (all of its ~40 draws) and shows the terrain it rendered; here the steps run on the worker
// thread through the C ABI and only the terrain hand-over needs the thread that owns the textures.
void HipTectonics::run()
{
    if (!hip_abi_matches(&error_)) { setProgress({ true, 1.0f }); return; }
    std::vector<float> img;
    dmg_grid g{};
    dispatchGPU([&](Project *p) {
        img = p->terrain;
        g.W = p->getWidth(); g.H = p->getHeight();
        for (int i = 0; i < 4; i++) g.coords[i] = p->coords[i];
        g.circumference = p->circumference;
    });
    dmg_ctx *ctx = nullptr;
    dmg_tectonics *t = nullptr;
    auto fail = [&](const char *where) {
        error_ = std::string(where) + ": " + dmg_last_error();
        if (t) dmg_tectonics_destroy(t);
        if (ctx) dmg_destroy(ctx);
        setProgress({ true, 1.0f });
    };
    if (!layer_matches(img, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; setProgress({ true, 1.0f }); return; }
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) return fail("dmg_create");
    if (dmg_tectonics_create(ctx, &g, img.data(), &t)) return fail("dmg_tectonics_create");     // Tectonics::Tectonics :15-154
    std::vector<float> terrain((size_t)g.W * g.H);
    for (int i = 0; i < n_steps_ && !cancelled(); i++) {                                                        // :157
        if (dmg_tectonics_step(t, 1)) return fail("dmg_tectonics_step");
        if (dmg_tectonics_get_terrain(t, terrain.data())) return fail("dmg_tectonics_get_terrain");
        dispatchGPU([&](Project *p) { p->terrain = terrain; });                                  // the render pass' target, :210
        setProgress({ false, (float)(i + 1) / (float)n_steps_ });
    }
    dmg_tectonics_destroy(t);
    dmg_destroy(ctx);
    setProgress({ true, 1.0f });                                                                // :274
}

// GradientNoiseFilter::step (reference src/filter/GradientNoise.cpp:61-500): one pass, finished at once
std::pair<bool, float> HipGradientNoise::step(Project *p)
{
    dmg_grid g{};
    g.W = p->getWidth(); g.H = p->getHeight();
    for (int i = 0; i < 4; i++) g.coords[i] = p->coords[i];
    g.circumference = p->circumference;
    if (!layer_matches(p->terrain, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; return { true, 1.0f }; }
    dmg_ctx *ctx = nullptr;
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) { error_ = std::string("dmg_create: ") + dmg_last_error(); return { true, 1.0f }; }
    // terrain := blend_mode(terrain, noise, selection); no selection layer in this mirror = all ones (src/Project.cpp:102-104)
    if (dmg_gradient_noise(ctx, &g, p->terrain.data(), nullptr, &params_)) error_ = std::string("dmg_gradient_noise: ") + dmg_last_error();
    dmg_destroy(ctx);
    return { true, 1.0f };                                                                       // :499
}
