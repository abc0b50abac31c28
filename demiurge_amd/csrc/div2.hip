// Exact division by a constant in two operations.
//
// The Jacobi shader divides three times per cell by numbers that depend on the row only (pixelwidth2.x, pixelwidth2.y and
// Beta, reference src/filter/OceanCurrents.cpp:478-480).  A correctly rounded quotient through the constant's reciprocal
// takes three operations in Markstein's form (dmg_div_rc).  Brisebarre and Muller ("Correctly rounded multiplication by
// arbitrary precision constants"): write 1/c = h + l with h = RN(1/c), l = RN(1/c - h); then
//
//     u = RN(x * l);   q = RN(x * h + u)          (one multiply, one fused multiply-add)
//
// is RN(x * (1/c)) = RN(x / c) for every x -- for most constants c, and for a given c this is decidable.  Here it is decided
// the blunt way, on the device: a floating-point quotient scales exactly with the exponent of x, so trying the 2^23
// significands of x in [1, 2) covers every x whose products neither underflow nor overflow.  For about one constant in a
// hundred the standard pair fails -- for a single significand; one of the neighbouring pairs (h one ulp up or down, l up
// to two ulps off the matching remainder) is then exact for all of them.  A pair is only ever handed out after all 2^23
// quotients have compared equal to the correctly rounded one, so whoever uses it computes IEEE quotients by construction.
//
// The test of one quotient uses no division: q is the correctly rounded x / c iff x - q c is no larger in magnitude than
// x - q' c for both neighbours q' of q; each remainder is exact in one fma (it has at most 24 significant bits when q is
// within an ulp of the quotient, and a q further off loses the comparison whatever the rounding), and a tie cannot occur
// (a quotient of two floats is never half-way between two floats).
//
// Window: as for dmg_div_rc the result is the IEEE quotient unless x * l underflows (|x| below ~2^-100 for the Jacobi
// constants) or the quotient overflows; x = +0 gives +0.  include/demiurge_hip.h states the window at dmg_ocean_jacobi.
#include "dmg_internal.h"
#include <mutex>
#include <unordered_map>

namespace {

__global__ __launch_bounds__(256) void k_div2_check(const float *c, const float *h, const float *l, unsigned *fails, int n)
{
    const int i = blockIdx.y;
    if (i >= n) return;
    const float cc = c[i], hh = h[i], ll = l[i];
    unsigned bad = 0;
    for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(0x3F800000u | m);
        const float q = __fmaf_rn(x, hh, __fmul_rn(x, ll));
        const float qu = __uint_as_float(__float_as_uint(q) + 1u), qd = __uint_as_float(__float_as_uint(q) - 1u);
        const float r0 = fabsf(__fmaf_rn(-q, cc, x)), ru = fabsf(__fmaf_rn(-qu, cc, x)), rd = fabsf(__fmaf_rn(-qd, cc, x));
        bad |= (r0 <= ru && r0 <= rd) ? 0u : 1u;              // (a NaN or an infinity anywhere counts as a failure)
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(&fails[i], 1u);
}

struct split { float h, l; bool ok; };
std::mutex g_mu;
std::unordered_map<unsigned, split> g_cache;                   // bit pattern of c -> its pair (a pure function of c)

// the k-th candidate pair of a constant: k = 0 the standard one, then h in {RN(1/c), one ulp down, one ulp up} and l up to
// two ulps off RN(1/c - h)
constexpr int NCAND = 15;
void candidate(float c, int k, float *h, float *l)
{
    const double r = 1.0 / (double)c;
    static const int dh_of[3] = { 0, -1, 1 }, dl_of[5] = { 0, -1, 1, -2, 2 };
    const int dh = dh_of[k / 5], dl = dl_of[k % 5];
    float hh = (float)r;
    if (dh) hh = nextafterf(hh, dh < 0 ? 0.0f : INFINITY);
    float ll = (float)(r - (double)hh);
    for (int s = 0; s < (dl < 0 ? -dl : dl); s++) ll = nextafterf(ll, dl < 0 ? -INFINITY : INFINITY);
    *h = hh; *l = ll;
}

bool usable(float c)                                           // positive, normal, with a normal reciprocal
{
    return c >= 0x1p-100f && c <= 0x1p100f;
}

// fails[i] != 0 where (h[i], l[i]) is not an exact pair of c[i]
int run_check(dmg_ctx *ctx, const std::vector<float> &c, const std::vector<float> &h, const std::vector<float> &l, std::vector<unsigned> &fails)
{
    const int n = (int)c.size();
    fails.assign((size_t)n, 0u);
    if (!n) return DMG_OK;
    float *d = nullptr;
    const size_t bytes = (size_t)n * 16;
    DMG_HIP(hipSetDevice(ctx->device));
    if (dmg_scratch_alloc(ctx, bytes, (void **)&d) != hipSuccess) return dmg_fail(DMG_ENOMEM, "scratch allocation of %zu bytes failed", bytes);
    float *dc = d, *dh = d + n, *dl = d + 2 * (size_t)n;
    unsigned *df = reinterpret_cast<unsigned *>(d + 3 * (size_t)n);
    hipStream_t st = ctx->stream;
    int rc = DMG_OK;
    if (hipMemcpyAsync(dc, c.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dh, h.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dl, l.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(df, 0, (size_t)n * 4, st) != hipSuccess)
        rc = dmg_fail(DMG_EHIP, "div2: upload failed");
    for (int i0 = 0; i0 < n && !rc; i0 += 32768) {
        const int m = std::min(32768, n - i0);
        // 16 blocks x 256 threads per constant: 2048 significands per thread
        hipLaunchKernelGGL(k_div2_check, dim3(16, (unsigned)m), dim3(256), 0, st, dc + i0, dh + i0, dl + i0, df + i0, m);
        if (hipGetLastError() != hipSuccess) rc = dmg_fail(DMG_EHIP, "div2: launch failed");
    }
    if (!rc && (hipMemcpyAsync(fails.data(), df, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
        rc = dmg_fail(DMG_EHIP, "div2: download failed");
    if (rc) (void)hipStreamSynchronize(st);
    dmg_scratch_free(ctx, d, bytes);
    return rc;
}

}  // namespace

// Exact pairs for n constants: x / c[i] == fma(x, h[i], x * l[i]) for every x inside the window.  ok[i] = 0 where no
// candidate pair is exact (or c[i] is not a positive normal number): the caller keeps the three-operation form there.
int dmg_div2_splits(dmg_ctx *ctx, const float *c, int n, float *h, float *l, unsigned char *ok)
{
    std::lock_guard<std::mutex> guard(g_mu);
    // constants this process has not seen yet, each once
    std::vector<float> todo;
    {
        std::unordered_map<unsigned, int> seen;
        for (int i = 0; i < n; i++) {
            unsigned bits;
            memcpy(&bits, &c[i], 4);
            if (!usable(c[i])) { g_cache[bits] = split{ 0.0f, 0.0f, false }; continue; }
            if (g_cache.count(bits) || seen.count(bits)) continue;
            seen[bits] = 1;
            todo.push_back(c[i]);
        }
    }
    if (!todo.empty()) {
        std::vector<float> hh(todo.size()), ll(todo.size());
        std::vector<unsigned> fails;
        for (size_t i = 0; i < todo.size(); i++) candidate(todo[i], 0, &hh[i], &ll[i]);
        int rc = run_check(ctx, todo, hh, ll, fails);
        if (rc) return rc;
        std::vector<float> again;
        for (size_t i = 0; i < todo.size(); i++) {
            unsigned bits;
            memcpy(&bits, &todo[i], 4);
            if (!fails[i]) g_cache[bits] = split{ hh[i], ll[i], true };
            else again.push_back(todo[i]);
        }
        if (!again.empty()) {
            std::vector<float> cc, ch, cl;
            for (float v : again)
                for (int k = 1; k < NCAND; k++) {
                    float a, b;
                    candidate(v, k, &a, &b);
                    cc.push_back(v); ch.push_back(a); cl.push_back(b);
                }
            rc = run_check(ctx, cc, ch, cl, fails);
            if (rc) return rc;
            for (size_t i = 0; i < again.size(); i++) {
                unsigned bits;
                memcpy(&bits, &again[i], 4);
                split s{ 0.0f, 0.0f, false };
                for (int k = 1; k < NCAND && !s.ok; k++) {
                    const size_t j = i * (NCAND - 1) + (size_t)(k - 1);
                    if (!fails[j]) s = split{ ch[j], cl[j], true };
                }
                g_cache[bits] = s;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        unsigned bits;
        memcpy(&bits, &c[i], 4);
        const split s = g_cache[bits];
        h[i] = s.h; l[i] = s.l; ok[i] = s.ok ? 1 : 0;
    }
    return DMG_OK;
}

extern "C" int dmg_div2_split(dmg_ctx *ctx, float c, float *h, float *l)
{
    DMG_CHECK_ARG(ctx != nullptr && h != nullptr && l != nullptr);
    unsigned char ok = 0;
    int rc = dmg_div2_splits(ctx, &c, 1, h, l, &ok);
    if (rc) return rc;
    if (!ok) return dmg_fail(DMG_EINVAL, "no exact two-operation pair for the divisor %a", (double)c);
    return DMG_OK;
}
