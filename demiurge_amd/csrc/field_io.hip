// Field export: the reference's PNG dump of the terrain (Project::file_write,
// src/Project.cpp:56-67) and the ocean-current display shading (src/appearance/VectorField.cpp:117-118) as data, so a
// run here can be laid beside the reference's misc/ocean_sim.png.  Host-side code: a PNG of stored (uncompressed)
// deflate blocks needs no zlib.
#include <cstdint>
#include "ocean_internal.h"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32_update(uint32_t c, const unsigned char *p, size_t n)
{
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t k = i;
            for (int j = 0; j < 8; j++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
            crc_table[i] = k;
        }
        crc_ready = true;
    }
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}

void put32(std::vector<unsigned char> &v, uint32_t x)
{ v.push_back(x >> 24); v.push_back((x >> 16) & 0xFF); v.push_back((x >> 8) & 0xFF); v.push_back(x & 0xFF); }

void chunk(std::vector<unsigned char> &out, const char *type, const std::vector<unsigned char> &data)
{
    put32(out, (uint32_t)data.size());
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc32_update(0xFFFFFFFFu, out.data() + at, out.size() - at) ^ 0xFFFFFFFFu);
}

// 8-bit greyscale PNG; rows in memory order (row 0 first), like stbi_write_png(..., comp = 1, stride = W)
int write_png_gray8(const char *path, int W, int H, const unsigned char *px)
{
    std::vector<unsigned char> raw;
    raw.reserve((size_t)H * (W + 1));
    for (int y = 0; y < H; y++) {
        raw.push_back(0);                                          // filter type: none
        raw.insert(raw.end(), px + (size_t)y * W, px + (size_t)(y + 1) * W);
    }
    std::vector<unsigned char> z = { 0x78, 0x01 };                 // zlib header, no compression
    uint32_t a = 1, b = 0;                                         // Adler-32
    for (size_t off = 0; off < raw.size();) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);                // BFINAL, BTYPE = 00 (stored)
        z.push_back(n & 0xFF); z.push_back(n >> 8);
        z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    put32(z, (b << 16) | a);
    std::vector<unsigned char> out = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)W); put32(ihdr, (uint32_t)H);
    ihdr.push_back(8); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit grey
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = fopen(path, "wb");
    if (!f) return dmg_fail(DMG_EINVAL, "cannot open %s for writing", path);
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok ? DMG_OK : dmg_fail(DMG_EINVAL, "short write to %s", path);
}

// VectorField.cpp:117-118: vel = (scratch1, scratch2); if (img <= 0) fc = 1 - length(vel)*8   (land keeps `land`)
__global__ void k_speed_shade(const float *__restrict__ img, const float2 *__restrict__ v, float *__restrict__ out, size_t n, float land)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 w = v[i];
    float len = sqrtf(__fadd_rn(__fmul_rn(w.x, w.x), __fmul_rn(w.y, w.y)));
    out[i] = img[i] <= 0.0f ? __fsub_rn(1.0f, __fmul_rn(len, 8.0f)) : land;
}

}  // namespace

extern "C" {

int dmg_field_write_png(const char *path, int W, int H, const float *field, int clamp)
{
    DMG_CHECK_ARG(path != nullptr && field != nullptr && W > 0 && H > 0);
    std::vector<unsigned char> px((size_t)W * H);
    for (size_t i = 0; i < px.size(); i++) {
        float f = field[i];
        if (clamp) f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
        else if (!(f > -8.0e6f && f < 8.0e6f)) f = 0.0f;           // int conversion of the reference is undefined out here
        int a = (int)(255.0f * f);                                 // Project.cpp:60-61: int a = 255.0f*image[i]; data[i] = a;
        px[i] = (unsigned char)a;
    }
    return write_png_gray8(path, W, H, px.data());
}

int dmg_ocean_speed_shade(dmg_ocean *o, float land_value, float *out_full)
{
    DMG_CHECK_ARG(o != nullptr && out_full != nullptr);
    if (!o->have_terrain) return dmg_fail(DMG_ESTATE, "dmg_ocean_set_terrain must come first");
    size_t n = (size_t)o->L * o->grid.W;
    float *d = nullptr;
    DMG_HIP(hipMalloc((void **)&d, n * sizeof(float)));
    hipLaunchKernelGGL(k_speed_shade, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, o->ctx->stream, o->img,
                       reinterpret_cast<const float2 *>(o->v[o->vcur]), d, n, land_value);
    int rc = ocean_download_rows(o, out_full, d, 1);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
