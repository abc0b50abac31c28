// Checkpoint / resume of an ocean run as NumPy .npy files (the reference has no checkpointing;
// its ocean solve runs 10000 outer steps, reference src/filter/OceanCurrents.cpp:155).  One set of files per band:
//   <prefix>.v.npy         float32 (rows, W, 2)   velocity, interleaved like the reference's RG32F texture
//   <prefix>.pressure.npy  float32 (rows, W)
//   <prefix>.divw.npy      float32 (rows, W)
//   <prefix>.meta          text sidecar: grid, band, params and counters; written LAST, so its presence marks a complete set
// rows = the band's OWNED rows; ghost rows are refreshed by the next halo exchange.  Every file is written under a
// temporary name and renamed into place, so a crash mid-checkpoint leaves the previous set intact.
#include "ocean_internal.h"
#include <unistd.h>

static int commit_file(FILE *f, bool ok, const std::string &tmp, const std::string &path)
{
    ok = (fflush(f) == 0) && ok;
    ok = (fsync(fileno(f)) == 0) && ok;
    ok = (fclose(f) == 0) && ok;
    if (ok && rename(tmp.c_str(), path.c_str()) != 0) ok = false;
    if (!ok) { (void)remove(tmp.c_str()); return dmg_fail(DMG_EINVAL, "short write to %s", path.c_str()); }
    return DMG_OK;
}

static int write_npy(const std::string &path, const float *data, int rows, int W, int comps)
{
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return dmg_fail(DMG_EINVAL, "cannot open %s for writing", tmp.c_str());
    char dict[128];
    if (comps == 1) snprintf(dict, sizeof dict, "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d), }", rows, W);
    else snprintf(dict, sizeof dict, "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d, %d), }", rows, W, comps);
    std::string h(dict);
    size_t total = 10 + h.size() + 1;                 // magic(6) + version(2) + len(2) + dict + '\n'
    size_t pad = (64 - total % 64) % 64;
    h.append(pad, ' ');
    h.push_back('\n');
    unsigned short hl = (unsigned short)h.size();
    const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(&hl, 2, 1, f) == 1 && fwrite(h.data(), 1, h.size(), f) == h.size();
    size_t n = (size_t)rows * W * comps;
    ok = ok && fwrite(data, sizeof(float), n, f) == n;
    return commit_file(f, ok, tmp, path);
}

// what a restore must agree with: the canvas, the band split and the solver parameters
static std::string meta_text(const dmg_ocean *o, bool counters)
{
    char b[1024];
    const dmg_grid &g = o->grid;
    const dmg_ocean_params &q = o->prm;
    int n = snprintf(b, sizeof b,
                     "demiurge_hip ocean checkpoint v1\nW %d H %d\ncoords %a %a %a %a\ncircumference %a\n"
                     "band %d of %d rows %d %d\nparams pressurefactor %a jacobi_iters %d diffusion_iters %d timestep_h %a scale %a wind %a quirks %d\n",
                     g.W, g.H, g.coords[0], g.coords[1], g.coords[2], g.coords[3], g.circumference, o->rank, o->count, o->y0, o->y1,
                     q.pressurefactor, q.jacobi_iters, q.diffusion_iters, q.timestep_h, q.scale, q.wind, q.quirks);
    std::string s(b, (size_t)n);
    if (counters) {
        snprintf(b, sizeof b, "counters jacobi_launches %lld jacobi_sweeps %lld outer_steps %lld\n", o->n_launches, o->n_sweeps, o->n_outer);
        s += b;
    }
    return s;
}

static int read_npy(const std::string &path, float *data, int rows, int W, int comps)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return dmg_fail(DMG_EINVAL, "cannot open %s", path.c_str());
    unsigned char head[10];
    if (fread(head, 1, 10, f) != 10 || memcmp(head, "\x93NUMPY", 6) != 0 || head[6] != 1) {
        fclose(f);
        return dmg_fail(DMG_EINVAL, "%s is not a version-1 .npy file", path.c_str());
    }
    unsigned hl = head[8] | (head[9] << 8);
    std::string dict(hl, ' ');
    bool ok = fread(&dict[0], 1, hl, f) == hl;
    char want[64];
    if (comps == 1) snprintf(want, sizeof want, "'shape': (%d, %d)", rows, W);
    else snprintf(want, sizeof want, "'shape': (%d, %d, %d)", rows, W, comps);
    if (!ok || dict.find("'<f4'") == std::string::npos || dict.find(want) == std::string::npos ||
        dict.find("'fortran_order': False") == std::string::npos) {
        fclose(f);
        return dmg_fail(DMG_EINVAL, "%s: expected little-endian float32 C-order with %s, header is %s", path.c_str(), want, dict.c_str());
    }
    size_t n = (size_t)rows * W * comps;
    ok = fread(data, sizeof(float), n, f) == n;
    fclose(f);
    return ok ? DMG_OK : dmg_fail(DMG_EINVAL, "%s is truncated", path.c_str());
}

extern "C" {

int dmg_ocean_checkpoint(dmg_ocean *o, const char *prefix)
{
    DMG_CHECK_ARG(o != nullptr && prefix != nullptr);
    int rc = no_pending(o, "dmg_ocean_checkpoint");        // the pressure buffers are half a launch apart in between
    if (rc) return rc;
    DMG_HIP(hipSetDevice(o->ctx->device));
    const int own = o->y1 - o->y0, W = o->grid.W;
    const size_t n = (size_t)own * W, off = (size_t)o->gs * W;
    std::vector<float> buf(n * 2);
    hipStream_t st = o->ctx->stream;
    std::string p(prefix);
    (void)remove((p + ".meta").c_str());                   // the set is incomplete until the new sidecar is in place
    DMG_HIP(hipMemcpyAsync(buf.data(), o->v[o->vcur] + off * 2, n * 8, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = write_npy(p + ".v.npy", buf.data(), own, W, 2))) return rc;
    DMG_HIP(hipMemcpyAsync(buf.data(), o->p[o->cur] + off, n * 4, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = write_npy(p + ".pressure.npy", buf.data(), own, W, 1))) return rc;
    DMG_HIP(hipMemcpyAsync(buf.data(), o->divw + off, n * 4, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = write_npy(p + ".divw.npy", buf.data(), own, W, 1))) return rc;
    const std::string meta = meta_text(o, true), mp = p + ".meta", tmp = mp + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return dmg_fail(DMG_EINVAL, "cannot open %s for writing", tmp.c_str());
    return commit_file(f, fwrite(meta.data(), 1, meta.size(), f) == meta.size(), tmp, mp);
}

int dmg_ocean_restore(dmg_ocean *o, const char *prefix)
{
    DMG_CHECK_ARG(o != nullptr && prefix != nullptr);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_restore: dmg_ocean_set_terrain must come first");
    int rc = no_pending(o, "dmg_ocean_restore");
    if (rc) return rc;
    DMG_HIP(hipSetDevice(o->ctx->device));
    const int own = o->y1 - o->y0, W = o->grid.W;
    const size_t n = (size_t)own * W, off = (size_t)o->gs * W;
    std::vector<float> buf(n * 2);
    hipStream_t st = o->ctx->stream;
    std::string p(prefix);
    {                                                      // the sidecar must describe this canvas, band and parameter set
        FILE *f = fopen((p + ".meta").c_str(), "rb");
        if (!f) return dmg_fail(DMG_EINVAL, "%s.meta is missing: not a complete checkpoint", prefix);
        std::string got(4096, '\0');
        got.resize(fread(&got[0], 1, got.size(), f));
        fclose(f);
        const std::string want = meta_text(o, false);
        if (got.compare(0, want.size(), want) != 0)
            return dmg_fail(DMG_EINVAL, "%s.meta describes another run:\n%swanted:\n%s", prefix, got.c_str(), want.c_str());
        long long a = 0, b = 0, c = 0;
        if (sscanf(got.c_str() + want.size(), "counters jacobi_launches %lld jacobi_sweeps %lld outer_steps %lld", &a, &b, &c) == 3)
            o->n_outer = c;                                // the step index resumes; launch statistics stay this process's
    }
    if ((rc = read_npy(p + ".v.npy", buf.data(), own, W, 2))) return rc;
    DMG_HIP(hipMemcpyAsync(o->v[o->vcur] + off * 2, buf.data(), n * 8, hipMemcpyHostToDevice, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = read_npy(p + ".pressure.npy", buf.data(), own, W, 1))) return rc;
    DMG_HIP(hipMemcpyAsync(o->p[o->cur] + off, buf.data(), n * 4, hipMemcpyHostToDevice, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = read_npy(p + ".divw.npy", buf.data(), own, W, 1))) return rc;
    DMG_HIP(hipMemcpyAsync(o->divw + off, buf.data(), n * 4, hipMemcpyHostToDevice, st));
    DMG_HIP(hipStreamSynchronize(st));
    return ocean_zero_land(o);                       // whatever the file held: land keeps exactly +0 in both pressure buffers
}

}  // extern "C"
