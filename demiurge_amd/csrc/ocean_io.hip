// Checkpoint / resume of an ocean run as NumPy .npy files (the reference has no checkpointing;
// its ocean solve runs 10000 outer steps, reference src/filter/OceanCurrents.cpp:155).  One set of files per band and
// generation N (1, 2, ...):
//   <prefix>.gN.v.npy         float32 (rows, W, 2)   velocity, interleaved like the reference's RG32F texture
//   <prefix>.gN.pressure.npy  float32 (rows, W)
//   <prefix>.gN.divw.npy      float32 (rows, W)
//   <prefix>.meta             text sidecar: the generation it points at, grid, band, params and counters
// rows = the band's OWNED rows; ghost rows are refreshed by the next halo exchange.
// Atomicity: a checkpoint writes a NEW generation next to the one the sidecar points at, syncs the three
// arrays, then replaces the sidecar with one rename -- the only step that changes what a restore sees -- syncs the
// directory and only then removes the previous generation.  A crash at any point leaves a sidecar that points at a
// complete set: the old one before the rename, the new one after it.
#include "ocean_internal.h"
#include <fcntl.h>
#include <unistd.h>

static void sync_dir_of(const std::string &path)
{
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : path.substr(0, slash));
    const int fd = open(dir.c_str(), O_RDONLY | O_DIRECTORY);
    if (fd >= 0) { (void)fsync(fd); (void)close(fd); }
}

// the generation the sidecar of `prefix` points at (0: no usable sidecar) and its text
static int read_meta(const std::string &prefix, std::string *text)
{
    FILE *f = fopen((prefix + ".meta").c_str(), "rb");
    if (!f) return 0;
    std::string got(4096, '\0');
    got.resize(fread(&got[0], 1, got.size(), f));
    fclose(f);
    int gen = 0;
    const size_t at = got.find("generation ");
    if (at == std::string::npos || sscanf(got.c_str() + at, "generation %d", &gen) != 1 || gen < 1) return 0;
    if (text) *text = got;
    return gen;
}

static std::string gen_name(const std::string &prefix, int gen, const char *what)
{
    return prefix + ".g" + std::to_string(gen) + "." + what + ".npy";
}

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
// (quirks without round 5's DMG_QUIRK_EXACT_LLVMPIPE: that bit was set by default then and means nothing now -- a sidecar written
// by a round-5 default handle, "quirks 5", describes the same run as today's default "quirks 1"; meta_normalised does the same to
// the text read back)
static std::string meta_normalised(const std::string &got)
{
    const size_t k = got.find(" quirks ");
    if (k == std::string::npos) return got;
    const size_t v0 = k + 8, v1 = got.find('\n', v0);
    if (v1 == std::string::npos) return got;
    char *end = nullptr;
    const long q = strtol(got.c_str() + v0, &end, 10);
    if (end != got.c_str() + v1) return got;
    return got.substr(0, v0) + std::to_string((int)q & ~DMG_QUIRK_EXACT_LLVMPIPE) + got.substr(v1);
}
static std::string meta_text(const dmg_ocean *o, bool counters)
{
    char b[1024];
    const dmg_grid &g = o->grid;
    const dmg_ocean_params &q = o->prm;
    int n = snprintf(b, sizeof b,
                     "demiurge_hip ocean checkpoint v2\nW %d H %d\ncoords %a %a %a %a\ncircumference %a\n"
                     "band %d of %d rows %d %d\nparams pressurefactor %a jacobi_iters %d diffusion_iters %d timestep_h %a scale %a wind %a quirks %d\n",
                     g.W, g.H, g.coords[0], g.coords[1], g.coords[2], g.coords[3], g.circumference, o->rank, o->count, o->y0, o->y1,
                     q.pressurefactor, q.jacobi_iters, q.diffusion_iters, q.timestep_h, q.scale, q.wind, q.quirks & ~DMG_QUIRK_EXACT_LLVMPIPE);
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
    const std::string p(prefix);
    const int old_gen = read_meta(p, nullptr), gen = old_gen + 1;
    // the new generation, next to the one the sidecar points at (which stays untouched until the sidecar is replaced)
    DMG_HIP(hipMemcpyAsync(buf.data(), o->v[o->vcur] + off * 2, n * 8, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = write_npy(gen_name(p, gen, "v"), buf.data(), own, W, 2))) return rc;
    DMG_HIP(hipMemcpyAsync(buf.data(), o->p[o->cur] + off, n * 4, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = write_npy(gen_name(p, gen, "pressure"), buf.data(), own, W, 1))) return rc;
    DMG_HIP(hipMemcpyAsync(buf.data(), o->divw + off, n * 4, hipMemcpyDeviceToHost, st));
    DMG_HIP(hipStreamSynchronize(st));
    if ((rc = write_npy(gen_name(p, gen, "divw"), buf.data(), own, W, 1))) return rc;
    sync_dir_of(p);                                        // the three names exist on disk before the sidecar says so
    // the flip: one rename of the sidecar
    char g[64];
    snprintf(g, sizeof g, "generation %d\n", gen);
    const std::string meta = meta_text(o, true) + g, mp = p + ".meta", tmp = mp + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return dmg_fail(DMG_EINVAL, "cannot open %s for writing", tmp.c_str());
    if ((rc = commit_file(f, fwrite(meta.data(), 1, meta.size(), f) == meta.size(), tmp, mp))) return rc;
    sync_dir_of(p);
    if (old_gen > 0)                                       // nothing points at the previous generation any more
        for (const char *what : { "v", "pressure", "divw" }) (void)remove(gen_name(p, old_gen, what).c_str());
    return DMG_OK;
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
    hipStream_t st = o->ctx->stream;
    const std::string p(prefix);
    std::string got;
    const int gen = read_meta(p, &got);
    if (gen < 1) return dmg_fail(DMG_EINVAL, "%s.meta is missing or names no generation: not a complete checkpoint", prefix);
    long long outer = o->n_outer;
    {                                                      // the sidecar must describe this canvas, band and parameter set
        const std::string want = meta_text(o, false);
        got = meta_normalised(got);
        if (got.compare(0, want.size(), want) != 0)
            return dmg_fail(DMG_EINVAL, "%s.meta describes another run:\n%swanted:\n%s", prefix, got.c_str(), want.c_str());
        long long a = 0, b = 0, c = 0;
        if (sscanf(got.c_str() + want.size(), "counters jacobi_launches %lld jacobi_sweeps %lld outer_steps %lld", &a, &b, &c) == 3) outer = c;
    }
    // all three arrays are read and checked BEFORE anything of the handle changes: a failed restore leaves it as it was
    std::vector<float> v(n * 2), pr(n), dw(n);
    if ((rc = read_npy(gen_name(p, gen, "v"), v.data(), own, W, 2))) return rc;
    if ((rc = read_npy(gen_name(p, gen, "pressure"), pr.data(), own, W, 1))) return rc;
    if ((rc = read_npy(gen_name(p, gen, "divw"), dw.data(), own, W, 1))) return rc;
    DMG_HIP(hipMemcpyAsync(o->v[o->vcur] + off * 2, v.data(), n * 8, hipMemcpyHostToDevice, st));
    DMG_HIP(hipMemcpyAsync(o->p[o->cur] + off, pr.data(), n * 4, hipMemcpyHostToDevice, st));
    DMG_HIP(hipMemcpyAsync(o->divw + off, dw.data(), n * 4, hipMemcpyHostToDevice, st));
    DMG_HIP(hipStreamSynchronize(st));
    o->n_outer = outer;                                    // the step index resumes; launch statistics stay this process's
    return ocean_zero_land(o);                       // whatever the file held: land keeps exactly +0 in both pressure buffers
}

}  // extern "C"
