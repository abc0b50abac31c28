// dmg_world: BASELINE.json config 5 -- tectonic uplift, stream-power erosion and the year-cycle climate on one terrain.
//
// The reference has no coupled driver: its tools are applied one after the other to the shared terrain layer by the
// user (Project::get_terrain(), src/Project.h).  This handle is that sequence, kept on the device side of the C ABI:
//     Tectonics::run steps (src/filter/tectonics/Tectonics.cpp:156-275)   plates move, the map is rendered into the terrain
//  -> cpufilter::run rounds (src/filter/cpufilter.cpp:41-222)             uplift = max(h,0)/N of that terrain, FlowFilter + erosion pass
//  -> Temperature::run steps (src/filter/Temperature.cpp:47-162)          the energy-balance model over the eroded land/sea mask,
//                                                                         its orbital phase M advancing step by step through the year
// One-way coupling, as in the editor: erosion does not feed back into the plates (Tectonics rebuilds its plates from a
// terrain only in its constructor, :15-58), the climate sees the eroded terrain's sign.
//
// Round 3: the terrain never leaves the device between the stages (four 512 MiB PCIe crossings per iteration
// at 16384x8192).  Tectonics renders into its own device buffer, the erosion handle is created from that buffer with a
// device copy, the climate takes the eroded heights with a device copy; the reference keeps all of it in textures
// (cpufilter.cpp:106-216).  dmg_world_get downloads what the host asks for.
#include <cstring>
#include "dmg_internal.h"
#include <memory>

struct dmg_world {
    dmg_ctx *ctx = nullptr;
    dmg_grid grid{};
    dmg_tectonics *tect = nullptr;
    dmg_climate *clim = nullptr;
    float *d_terrain = nullptr, *d_eroded = nullptr; // what the tectonics render pass left; after the erosion rounds (scaled by 1/N like the reference)
    long long climate_step = 0, iterations = 0;
    dmg_climate_params clim_prm{};                   // what the climate handle holds (handed on only when a step brings other values)
};

extern "C" {

void dmg_world_params_default(dmg_world_params *p)
{
    if (!p) return;
    p->tectonics_steps = 1;
    p->erosion_rounds = 1;
    p->climate_steps = 10;                           // Temperature::run steps the field ten times per frame (:140)
    p->erosion_N = 50;                               // cpufilter.cpp:42
    p->exponent = 0.5f; p->slope_exponent = 1.0f; p->factor = 1.0f;      // the menu's defaults, cpufilter.h:18-21
    p->lakes = 0;
    dmg_climate_params_default(&p->climate);
}

int dmg_world_create(dmg_ctx *ctx, const dmg_grid *grid, const float *terrain, const float *T0, dmg_world **out)
{
    DMG_CHECK_ARG(ctx != nullptr && grid != nullptr && terrain != nullptr && T0 != nullptr && out != nullptr);
    *out = nullptr;
    std::unique_ptr<dmg_world> w(new (std::nothrow) dmg_world);
    if (!w) return dmg_fail(DMG_ENOMEM, "out of host memory");
    w->ctx = ctx; w->grid = *grid;
    const size_t n = (size_t)grid->W * grid->H;
    DMG_HIP(hipSetDevice(ctx->device));
    if (hipMalloc((void **)&w->d_terrain, n * 4) != hipSuccess || hipMalloc((void **)&w->d_eroded, n * 4) != hipSuccess) {
        (void)hipFree(w->d_terrain);
        return dmg_fail(DMG_ENOMEM, "hipMalloc of two %zu-cell fields failed", n);
    }
    int rc = DMG_OK;
    if (hipMemcpyAsync(w->d_terrain, terrain, n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(w->d_eroded, w->d_terrain, n * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = dmg_fail(DMG_EHIP, "terrain upload failed");
    if (!rc) rc = dmg_tectonics_create(ctx, grid, terrain, &w->tect);
    if (!rc) rc = dmg_climate_create(ctx, grid, T0, terrain, &w->clim);
    if (rc) {
        if (w->tect) dmg_tectonics_destroy(w->tect);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(w->d_terrain); (void)hipFree(w->d_eroded);
        return rc;
    }
    dmg_climate_params_default(&w->clim_prm);
    dmg_ctx_retain(ctx);
    *out = w.release();
    return DMG_OK;
}

void dmg_world_destroy(dmg_world *w)
{
    if (!w) return;
    dmg_ctx *ctx = w->ctx;
    if (w->clim) dmg_climate_destroy(w->clim);
    if (w->tect) dmg_tectonics_destroy(w->tect);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(w->d_terrain); (void)hipFree(w->d_eroded);
    delete w;
    dmg_ctx_release(ctx);
}

int dmg_world_step(dmg_world *w, const dmg_world_params *params)
{
    DMG_CHECK_ARG(w != nullptr);
    dmg_world_params p;
    if (params) p = *params; else dmg_world_params_default(&p);
    DMG_CHECK_ARG(p.tectonics_steps >= 0 && p.erosion_rounds >= 0 && p.climate_steps >= 0 && p.erosion_N >= 1);
    DMG_HIP(hipSetDevice(w->ctx->device));
    hipStream_t st = w->ctx->stream;
    const size_t bytes = (size_t)w->grid.W * w->grid.H * 4;
    int rc;
    if (p.tectonics_steps > 0) {
        if ((rc = dmg_tectonics_step(w->tect, p.tectonics_steps))) return rc;
        DMG_HIP(hipMemcpyAsync(w->d_terrain, tectonics_terrain_device(w->tect), bytes, hipMemcpyDeviceToDevice, st));
    }
    if (p.erosion_rounds > 0) {
        dmg_erosion *e = nullptr;
        if ((rc = erosion_create_from(w->ctx, &w->grid, w->d_terrain, true, p.erosion_N, p.exponent, p.slope_exponent, p.factor, p.lakes, &e))) return rc;
        rc = dmg_erosion_rounds(e, p.erosion_rounds);
        if (!rc && hipMemcpyAsync(w->d_eroded, erosion_heights_device(e), bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
            rc = dmg_fail(DMG_EHIP, "device copy of the eroded heights failed");
        dmg_erosion_destroy(e);                          // (its buffers go back to the context's pool; the stream orders their reuse)
        if (rc) return rc;
    } else DMG_HIP(hipMemcpyAsync(w->d_eroded, w->d_terrain, bytes, hipMemcpyDeviceToDevice, st));
    if (p.climate_steps > 0) {
        if (memcmp(&p.climate, &w->clim_prm, sizeof p.climate) != 0) {
            if ((rc = dmg_climate_set_params(w->clim, &p.climate))) return rc;
            w->clim_prm = p.climate;
        }
        if ((rc = climate_set_terrain_device(w->clim, w->d_eroded))) return rc;
        if ((rc = dmg_climate_steps(w->clim, (int)w->climate_step, p.climate_steps))) return rc;
        w->climate_step += p.climate_steps;
    }
    w->iterations++;
    return DMG_OK;
}

int dmg_world_get(dmg_world *w, float *terrain, float *eroded, float *T)
{
    DMG_CHECK_ARG(w != nullptr);
    DMG_HIP(hipSetDevice(w->ctx->device));
    const size_t bytes = (size_t)w->grid.W * w->grid.H * 4;
    if (terrain) DMG_HIP(hipMemcpyAsync(terrain, w->d_terrain, bytes, hipMemcpyDeviceToHost, w->ctx->stream));
    if (eroded) DMG_HIP(hipMemcpyAsync(eroded, w->d_eroded, bytes, hipMemcpyDeviceToHost, w->ctx->stream));
    DMG_HIP(hipStreamSynchronize(w->ctx->stream));
    if (T) return dmg_climate_get(w->clim, T);
    return DMG_OK;
}

}  // extern "C"
