#include "HipOceanCurrents.h"
#include "HipAbi.h"
#include <memory>

// OceanCurrents::run (reference src/filter/OceanCurrents.cpp:151-223) with the GL passes replaced by the C ABI.
// Only data hand-over needs the thread that owns the textures; compute runs on this worker thread.
void HipOceanCurrents::run()
{
    if (!hip_abi_matches(&error_)) { setProgress({ true, 1.0f }); return; }
    std::vector<float> img;
    dmg_grid g{};
    dispatchGPU([&](Project *p) {                       // Texture::downloadDataRAW of the terrain layer
        img = p->terrain;
        g.W = p->getWidth(); g.H = p->getHeight();
        for (int i = 0; i < 4; i++) g.coords[i] = p->coords[i];
        g.circumference = p->circumference;
    });

    // dmg_multi owns the bands, their streams and threads and every halo exchange (RCCL over xGMI between GPUs, peer
    // copies between bands that share one); with one device it is a plain dmg_ocean underneath
    dmg_multi *oc = nullptr;
    auto fail = [&](const char *where) {
        error_ = std::string(where) + ": " + dmg_last_error();
        if (oc) dmg_multi_destroy(oc);
        setProgress({ true, 1.0f });
    };
    if (!layer_matches(img, g.W, g.H)) { error_ = "terrain layer does not hold width*height texels"; setProgress({ true, 1.0f }); return; }
    dmg_ocean_params prm;
    dmg_ocean_params_default(&prm);
    prm.pressurefactor = pressure_;
    prm.jacobi_iters = jacobi_iters_;
    prm.diffusion_iters = diffusion_iters_;
    if (dmg_multi_create((int)devices_.size(), devices_.data(), &g, &prm, ghost_, DMG_MULTI_AUTO, &oc)) return fail("dmg_multi_create");
    if (dmg_multi_set_terrain(oc, img.data())) return fail("dmg_multi_set_terrain");

    std::vector<float> vx((size_t)g.W * g.H), vy((size_t)g.W * g.H);
    for (int j = 0; j < n_outer_ && !cancelled(); j++) {   // :155; Cancel (Filter.cpp:105-108) ends the solve at the next outer step
        if (dmg_multi_step(oc, 1)) return fail("dmg_multi_step");                         // advect .. subDiv (:156-168)
        if (dmg_multi_get(oc, vx.data(), vy.data(), nullptr, nullptr)) return fail("dmg_multi_get");
        dispatchGPU([&](Project *p) {                   // :175-205  scratch1 := v.x, scratch2 := v.y
            p->scratch1 = vx;
            p->scratch2 = vy;
        });
        setProgress({ false, (float)(j + 1) / (float)n_outer_ });
    }
    dmg_multi_destroy(oc);
    setProgress({ true, 1.0f });
}
