#include "HipOceanCurrents.h"
#include <memory>

// OceanCurrents::run (reference src/filter/OceanCurrents.cpp:151-223) with the GL passes replaced by the C ABI.
// Only data hand-over needs the thread that owns the textures; compute runs on this worker thread.
void HipOceanCurrents::run()
{
    std::vector<float> img;
    dmg_grid g{};
    dispatchGPU([&](Project *p) {                       // Texture::downloadDataRAW of the terrain layer
        img = p->terrain;
        g.W = p->getWidth(); g.H = p->getHeight();
        for (int i = 0; i < 4; i++) g.coords[i] = p->coords[i];
        g.circumference = p->circumference;
    });

    dmg_ctx *ctx = nullptr;
    dmg_ocean *oc = nullptr;
    auto fail = [&](const char *where) {
        error_ = std::string(where) + ": " + dmg_last_error();
        if (oc) dmg_ocean_destroy(oc);
        if (ctx) dmg_destroy(ctx);
        setProgress({ true, 1.0f });
    };
    if (!layer_matches(img, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; setProgress({ true, 1.0f }); return; }
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) return fail("dmg_create");
    dmg_ocean_params prm;
    dmg_ocean_params_default(&prm);
    prm.pressurefactor = pressure_;
    prm.jacobi_iters = jacobi_iters_;
    prm.diffusion_iters = diffusion_iters_;
    if (dmg_ocean_create(ctx, &g, &prm, 0, 1, 1, &oc)) return fail("dmg_ocean_create");
    if (dmg_ocean_set_terrain(oc, img.data())) return fail("dmg_ocean_set_terrain");

    std::vector<float> vx((size_t)g.W * g.H), vy((size_t)g.W * g.H);
    for (int j = 0; j < n_outer_; j++) {                // :155
        if (dmg_ocean_step(oc, 1)) return fail("dmg_ocean_step");                       // advect .. subDiv (:156-168)
        if (dmg_ocean_get(oc, vx.data(), vy.data(), nullptr, nullptr)) return fail("dmg_ocean_get");
        dispatchGPU([&](Project *p) {                   // :175-205  scratch1 := v.x, scratch2 := v.y
            p->scratch1 = vx;
            p->scratch2 = vy;
        });
        setProgress({ false, (float)(j + 1) / (float)n_outer_ });
    }
    dmg_ocean_destroy(oc);
    dmg_destroy(ctx);
    setProgress({ true, 1.0f });
}
