// Drives HipOceanCurrents the way the editor does: ProgressFilter::run calls subFilter->step(p) once per frame
// until it reports finished (reference src/filter/Filter.cpp:97-115).  Reads a terrain (raw fp32), writes vx, vy.
//   host_demo <W> <H> <terrain.f32> <n_outer> <jacobi_iters> <diffusion_iters> <out_prefix>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include "HipOceanCurrents.h"
#include "HipTectonics.h"
#include "HipSimulationTools.h"

static bool read_f32(const char *path, std::vector<float> &v)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    size_t n = fread(v.data(), sizeof(float), v.size(), f);
    fclose(f);
    return n == v.size();
}
static void write_f32(const std::string &path, const std::vector<float> &v)
{
    FILE *f = fopen(path.c_str(), "wb");
    fwrite(v.data(), sizeof(float), v.size(), f);
    fclose(f);
}

// host_demo tectonics <W> <H> <terrain.f32> <n_steps> <out_prefix>
// host_demo noise <W> <H> <mode 0..6> <seed> <out_prefix>
// host_demo erosion|climate|blur|dilate|erode|morphgrad|thermal|flow ...   (argument lists at each branch below)
static int other_tools(int argc, char **argv)
{
    std::string tool = argv[1];
    Project p;
    p.width = atoi(argv[2]); p.height = atoi(argv[3]);
    p.terrain.assign((size_t)p.width * p.height, 0.0f);
    std::unique_ptr<SubFilter> filter;
    std::string out;
    const std::string *err = nullptr;
    if (tool == "tectonics" && argc == 7) {
        if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
        auto f = std::make_unique<HipTectonics>(&p, atoi(argv[5]));
        err = &f->error(); filter = std::move(f); out = argv[6];
    } else if (tool == "noise" && argc == 7) {
        dmg_noise_params prm;
        dmg_noise_params_default(&prm);
        prm.mode = atoi(argv[4]); prm.seed = atoi(argv[5]);
        auto f = std::make_unique<HipGradientNoise>(prm);
        err = &f->error(); filter = std::move(f); out = argv[6];
    } else if (tool == "erosion" && argc == 8) {                 // host_demo erosion W H terrain.f32 rounds dolakes out
        if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
        auto f = std::make_unique<HipErosion>(&p, 0.5f, 1.0f, 1.0f, atoi(argv[6]), 50, atoi(argv[5]));      // the menu's defaults, cpufilter.h:18-21
        err = &f->error(); filter = std::move(f); out = argv[7];
    } else if (tool == "climate" && argc == 7) {                 // host_demo climate W H heights.f32 frames out
        if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
        auto f = std::make_unique<HipTemperature>(&p, atoi(argv[5]));
        err = &f->error(); filter = std::move(f); out = argv[6];
    } else if ((tool == "blur" || tool == "dilate" || tool == "erode" || tool == "morphgrad") && argc == 7) {   // host_demo blur W H field.f32 radius out
        if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
        std::unique_ptr<HipOneShot> f;
        if (tool == "blur") f = std::make_unique<HipBlur>(&p, (float)atof(argv[5]));
        else if (tool == "morphgrad") f = std::make_unique<HipMorphologicalGradient>(&p, (float)atof(argv[5]));
        else f = std::make_unique<HipMorphological>(&p, (float)atof(argv[5]), tool == "dilate" ? "max" : "min");
        err = &f->error(); filter = std::move(f); out = argv[6];
    } else if (tool == "thermal" && argc == 6) {                 // host_demo thermal W H terrain.f32 out
        if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
        auto f = std::make_unique<HipThermalErosion>();
        err = &f->error(); filter = std::move(f); out = argv[5];
    } else if (tool == "flow" && argc == 8) {                    // host_demo flow W H terrain.f32 exponent lakeflag out
        if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
        auto f = std::make_unique<HipFlow>(0.5f, (float)atof(argv[5]), atoi(argv[6]) != 0);
        err = &f->error(); filter = std::move(f); out = argv[7];
    } else { fprintf(stderr, "usage: host_demo {tectonics|noise|erosion|climate|blur|dilate|erode|morphgrad|thermal|flow} W H ... out (see host_demo.cpp)\n"); return 2; }
    int frames = 0;
    for (;;) {
        auto [finished, progress] = filter->step(&p);
        frames++;
        if (finished) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (!err->empty()) { fprintf(stderr, "filter failed: %s\n", err->c_str()); return 1; }
    write_f32(out + "_terrain.f32", p.terrain);
    printf("finished after %d frames\n", frames);
    return 0;
}

// host_demo lifetime: handles outlive their context.  The recorded round-1 abort was a dmg_ocean destroyed after
// dmg_destroy(ctx); the context now stays alive until its last handle goes (include/demiurge_hip.h, dmg_destroy).
static int lifetime_check()
{
    const int W = 256, H = 128;
    dmg_grid g = { W, H, { -1.57079632679f, 1.57079632679f, -3.14159265359f, 3.14159265359f }, 42000.0f };
    std::vector<float> terrain((size_t)W * H), field((size_t)W * H, 1.0f);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) terrain[(size_t)y * W + x] = ((x / 32 + y / 32) % 3 == 0) ? 5.0f : -5.0f;
    dmg_ctx *ctx = nullptr;
    if (dmg_create(0, nullptr, DMG_PRIVATE_STREAM, &ctx)) { fprintf(stderr, "dmg_create: %s\n", dmg_last_error()); return 1; }
    dmg_ocean *oc = nullptr; dmg_tectonics *te = nullptr; dmg_climate *cl = nullptr; dmg_erosion *er = nullptr;
    dmg_ocean_params prm; dmg_ocean_params_default(&prm);
    if (dmg_ocean_create(ctx, &g, &prm, 0, 1, 1, &oc) || dmg_tectonics_create(ctx, &g, terrain.data(), &te) ||
        dmg_climate_create(ctx, &g, field.data(), terrain.data(), &cl) ||
        dmg_erosion_create(ctx, &g, terrain.data(), 50, 0.5f, 1.0f, 1.0f, 0, &er)) { fprintf(stderr, "create: %s\n", dmg_last_error()); return 1; }
    dmg_destroy(ctx);                                    // wrong order on purpose: the handles keep the context alive
    int rc = dmg_ocean_set_terrain(oc, terrain.data());
    if (!rc) rc = dmg_ocean_set_divw(oc, field.data());
    if (!rc) rc = dmg_ocean_jacobi(oc, 16, 1);
    if (!rc) rc = dmg_ocean_get(oc, nullptr, nullptr, field.data(), nullptr);
    if (!rc) rc = dmg_tectonics_step(te, 1);
    if (!rc) rc = dmg_climate_steps(cl, 0, 2);
    if (!rc) rc = dmg_erosion_rounds(er, 1);
    if (rc) { fprintf(stderr, "work after dmg_destroy failed: %s\n", dmg_last_error()); return 1; }
    dmg_climate_destroy(cl);
    dmg_erosion_destroy(er);
    dmg_tectonics_destroy(te);
    dmg_ocean_destroy(oc);                               // the last handle releases the context
    printf("lifetime ok\n");
    return 0;
}

// Cancel (reference src/filter/Filter.cpp:105-108 sets `aborting` and restores the backup; its worker runs on): here the
// adapter's loop reads the flag once per outer step.  host_demo cancel W H terrain.f32: the reference's 10000 outer steps are
// asked for, Cancel arrives once the first has been shown, and the filter finishes within a step or two.
static int cancel_check(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: host_demo cancel W H terrain.f32\n"); return 2; }
    Project p;
    p.width = atoi(argv[2]); p.height = atoi(argv[3]);
    p.terrain.resize((size_t)p.width * p.height);
    if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
    auto filter = std::make_unique<HipOceanCurrents>(&p, 100.0f, 10000, 200, 10);
    int frames = 0, frames_at_cancel = -1;
    float progress_at_cancel = 0.0f, last = 0.0f;
    for (;;) {
        auto [finished, progress] = filter->step(&p);
        frames++;
        if (finished) break;
        last = progress;
        if (frames_at_cancel < 0 && progress > 0.0f) { filter->cancel(); frames_at_cancel = frames; progress_at_cancel = progress; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (!filter->error().empty()) { fprintf(stderr, "filter failed: %s\n", filter->error().c_str()); return 1; }
    printf("cancelled at outer step %d of 10000 (frame %d); finished at outer step %d (frame %d)\n", (int)(progress_at_cancel * 10000 + 0.5f),
           frames_at_cancel, (int)(last * 10000 + 0.5f), frames);
    // a filter destroyed in mid-run (the editor closing the modal): the destructor cancels, then joins
    auto second = std::make_unique<HipOceanCurrents>(&p, 100.0f, 10000, 200, 10);
    for (int i = 0; i < 50; i++) { second->step(&p); std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
    second.reset();
    printf("destroyed in mid-run\n");
    return 0;
}

// ProgressFilter / BackupFilter (reference src/filter/Filter.cpp:12-23,70-115): the tool as the editor runs it -- a backup of the target
// layer when the filter is made, one step per frame, and Cancel: `aborting` -> the backup is restored, the filter finalized.  HipErosion
// writes the terrain layer every round (cpufilter.cpp:200-216), so a cancel in mid-run finds a changed terrain to put back.
// host_demo progress W H terrain.f32
static int progress_check(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: host_demo progress W H terrain.f32\n"); return 2; }
    Project p;
    p.width = atoi(argv[2]); p.height = atoi(argv[3]);
    p.terrain.resize((size_t)p.width * p.height);
    if (!read_f32(argv[4], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
    const std::vector<float> before = p.terrain;
    auto target = [](Project *q) { return &q->terrain; };
    {   // cancelled in mid-run
        ProgressFilter f(&p, target, std::make_unique<HipErosion>(&p, 0.5f, 1.0f, 1.0f, 0, 50, 150));
        int frames = 0, changed_at = -1;
        while (!f.isFinished() && frames < 200000) {
            f.run(&p);
            frames++;
            if (changed_at < 0 && p.terrain != before) { changed_at = frames; f.abort(); }     // the first round's result has arrived: Cancel
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (!f.isFinished() || !f.wasAborted() || changed_at < 0) { fprintf(stderr, "FAIL: finished %d aborted %d changed_at %d\n", (int)f.isFinished(), (int)f.wasAborted(), changed_at); return 1; }
        if (p.terrain != before) { fprintf(stderr, "FAIL: the terrain is not the backup after Cancel\n"); return 1; }
        if (p.finalized != 1) { fprintf(stderr, "FAIL: finalizeFilter called %d times\n", p.finalized); return 1; }
        printf("cancelled at frame %d (progress %.3f): the terrain equals the backup, the filter is finalized\n", changed_at, f.getProgress());
    }   // (~ProgressFilter -> ~HipErosion: stop() joins the worker before the adapter's members go)
    {   // run to the end with half the canvas selected: the other half keeps the backup (restoreUnselected)
        p.selection.assign(p.terrain.size(), 0.0f);
        for (int y = 0; y < p.height; y++) for (int x = 0; x < p.width / 2; x++) p.selection[(size_t)y * p.width + x] = 1.0f;
        ProgressFilter f(&p, target, std::make_unique<HipErosion>(&p, 0.5f, 1.0f, 1.0f, 0, 50, 3));
        int frames = 0;
        while (!f.isFinished() && frames++ < 200000) { f.run(&p); std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
        if (!f.isFinished() || f.wasAborted()) { fprintf(stderr, "FAIL: the second filter did not finish\n"); return 1; }
        f.restoreUnselected();
        size_t kept = 0, moved = 0;
        for (int y = 0; y < p.height; y++)
            for (int x = 0; x < p.width; x++) {
                const size_t i = (size_t)y * p.width + x;
                if (x >= p.width / 2) kept += p.terrain[i] == before[i]; else moved += p.terrain[i] != before[i];
            }
        if (kept != (size_t)p.height * (p.width - p.width / 2) || moved == 0 || p.finalized != 2) { fprintf(stderr, "FAIL: kept %zu moved %zu finalized %d\n", kept, moved, p.finalized); return 1; }
        printf("finished: %zu selected texels changed, the %zu unselected ones hold the backup\n", moved, kept);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && std::string(argv[1]) == "progress") return progress_check(argc, argv);
    if (argc == 2 && std::string(argv[1]) == "lifetime") return lifetime_check();
    if (argc >= 2 && std::string(argv[1]) == "cancel") return cancel_check(argc, argv);
    if (argc >= 2 && !(argv[1][0] >= '0' && argv[1][0] <= '9')) return other_tools(argc, argv);
    if (argc != 8 && argc != 9) { fprintf(stderr, "usage: host_demo W H terrain.f32 n_outer jacobi_iters diffusion_iters out_prefix [n_bands on GPU 0]\n"); return 2; }
    Project p;
    p.width = atoi(argv[1]); p.height = atoi(argv[2]);
    p.terrain.resize((size_t)p.width * p.height);
    if (!read_f32(argv[3], p.terrain)) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
    std::vector<int> devices(argc == 9 ? (size_t)std::max(atoi(argv[8]), 1) : 1, 0);      // the rehearsal form: every band on GPU 0
    auto filter = std::make_unique<HipOceanCurrents>(&p, 100.0f, atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), devices,
                                                     std::min(32, p.height / (int)devices.size()));
    int frames = 0;
    for (;;) {                                           // one step() per UI frame
        auto [finished, progress] = filter->step(&p);
        frames++;
        if (finished) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (!filter->error().empty()) { fprintf(stderr, "filter failed: %s\n", filter->error().c_str()); return 1; }
    write_f32(std::string(argv[7]) + "_vx.f32", p.scratch1);
    write_f32(std::string(argv[7]) + "_vy.f32", p.scratch2);
    printf("finished after %d frames\n", frames);
    return 0;
}
