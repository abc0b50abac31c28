// GLSL transcendentals, evaluated the way the reference's pixels were produced.
//
// The reference runs its filters as GLSL; the parity target is the output of that GLSL (on Mesa llvmpipe, the only
// place it can run here).  Its sin/cos/tan/asin/acos/atan are not libm: Mesa's compiler expands asin, acos, atan and
// atan2 into short polynomials (abs. error ~1e-4 rad, allowed by the GLSL spec) and gallivm evaluates sin/cos with
// the Cephes octant scheme.  Wherever such a value selects a texel (tectonics' rotated gathers, advect's departure
// point), feeds a sum with 1e4 offsets (noise) or a row coefficient that 5000 Jacobi sweeps iterate on, "a few ulp
// from libm" is not the reference's result.  With these forms the Jacobi solve, divergence, subDiv, the climate step,
// tectonics, the noise generator and advect reproduce the reference's output bit for bit
// (tests/test_oracle_golden.py; the forms were identified with probe shaders, see oracle/tect_oracle.c).
//
// Usable on the host (row tables) and on the device.  Expressions of constants are folded by Mesa on the host with
// libm and stay libm calls at their call sites.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace glsl {

constexpr float PI_2 = 1.57079632679489661923f;
constexpr float PI_4 = 0.78539816339744830962f;

__host__ __device__ inline float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// sin and cos of one argument: octant reduction with a three-part pi/4, degree-3 minimax polynomials in z = x*x,
// fused multiply-adds where gallivm emits llvm.fmuladd
__host__ __device__ inline void sincos(float xin, float *sn, float *cs)
{
    float x = fabsf(xin);
    int j = (int)(x * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    x = fmaf(y, -0.78515625f, x);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = fmaf(yc, z, -1.388731625493765E-003f);
    yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = yc * z;
    yc = yc - z * 0.5f;
    yc = yc + 1.0f;
    float ys = -1.9515295891E-4f;
    ys = fmaf(ys, z, 8.3321608736E-3f);
    ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    int j2 = j - 2;
    float s = (j & 2) ? yc : ys, c = (j2 & 2) ? yc : ys;
    unsigned sb = __builtin_bit_cast(unsigned, s) ^ (__builtin_bit_cast(unsigned, xin) & 0x80000000u) ^ ((unsigned)(j & 4) << 29);
    unsigned cb = __builtin_bit_cast(unsigned, c) ^ ((unsigned)(~j2 & 4) << 29);
    *sn = __builtin_bit_cast(float, sb);
    *cs = __builtin_bit_cast(float, cb);
}
__host__ __device__ inline float sin(float x) { float s, c; sincos(x, &s, &c); return s; }
__host__ __device__ inline float cos(float x) { float s, c; sincos(x, &s, &c); return c; }
__host__ __device__ inline float tan(float x) { float s, c; sincos(x, &s, &c); return s / c; }

// sign(x) * (pi/2 - sqrt(1-|x|) * (pi/2 + |x| (pi/4 - 1 + |x| (p0 + |x| p1))))
__host__ __device__ inline float asin_poly(float x, float p0, float p1)
{
    float ax = fabsf(x);
    float t = p0 + ax * p1;
    t = (PI_4 - 1.0f) + ax * t;
    t = PI_2 + ax * t;
    return sign(x) * (PI_2 - sqrtf(1.0f - ax) * t);
}
__host__ __device__ inline float asin(float x) { return asin_poly(x, 0.086566724f, -0.03102955f); }
// acos: pi/2 - the same shape with its own two coefficients (Mesa's published acos expansion; 131072/131072 random
// arguments bit-equal to llvmpipe's output, with pi/4 - 1 taken in float as above)
__host__ __device__ inline float acos(float x) { return PI_2 - asin_poly(x, 0.08132463f, -0.02363318f); }

// atan of a >= 0: reduce to [0,1], odd degree-11 polynomial accumulated from the lowest power
__host__ __device__ inline float atan_pos(float a)
{
    float x = fminf(a, 1.0f) / fmaxf(a, 1.0f);
    float x2 = x * x, x3 = x2 * x, x5 = x3 * x2, x7 = x5 * x2, x9 = x7 * x2;
    float p = x * 0.9999793128310355f;
    p = p + x3 * -0.3326756418091246f;
    p = p + x5 * 0.1938924977115610f;
    p = p + x7 * -0.1173503194786851f;
    p = p + x9 * 0.0536813784310406f;
    p = p + (x9 * -0.0121323213173444f) * x2;   // x^11 is used once: Mesa multiplies x^9 by the constant first (probe: 131072/131072)
    return p + (a > 1.0f ? 1.0f : 0.0f) * (p * -2.0f + PI_2);
}
__host__ __device__ inline float atan(float x) { return atan_pos(fabsf(x)) * sign(x); }
__host__ __device__ inline float atan2(float y, float x)
{
    bool flip = 0.0f >= x;
    float s = flip ? fabsf(x) : y;
    float t = flip ? y : fabsf(x);
    float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
    float rcp_scaled_t = 1.0f / (t * scale);
    float s_over_t = (s * scale) * rcp_scaled_t;
    float tn = fabsf(x) == fabsf(y) ? 1.0f : fabsf(s_over_t);
    float arc = (flip ? 1.0f : 0.0f) * PI_2 + atan_pos(tn) * sign(tn);
    return fminf(y, rcp_scaled_t) < 0.0f ? -arc : arc;
}

// pow(x, y) = exp2(y * log2(x)) with gallivm's minimax polynomials (exp2: degree 5 on the fraction; log2: degree 4 in
// z = y*y, y = (m-1)/(m+1)), both as even/odd split Horner with fused multiply-adds.  Bit-equal to llvmpipe's exp2,
// log2 and pow on 16384/16384 random arguments; libm powf is 2e-6 away.
__host__ __device__ inline float exp2(float x)
{
    x = fminf(x, 128.0f);
    x = fmaxf(x, -126.99999f);
    float ip = floorf(x), fp = x - ip;
    float scale = __builtin_bit_cast(float, ((int)ip + 127) << 23);
    float x2 = fp * fp;
    float odd = 0.00187757667519147912699f, even = 0.00898934009049466391101f;
    odd = fmaf(x2, odd, 0.0558263180532956664775f); even = fmaf(x2, even, 0.240153617044375388211f);
    odd = fmaf(x2, odd, 0.693153073200168932794f); even = fmaf(x2, even, 1.000000000000000000000f);
    return scale * fmaf(odd, fp, even);
}
__host__ __device__ inline float log2(float x)
{
    int i = __builtin_bit_cast(int, x);
    float logexp = (float)(((i >> 23) & 255) - 127);
    float mant = __builtin_bit_cast(float, (i & 0x007fffff) | 0x3f800000);
    float y = (mant - 1.0f) / (mant + 1.0f);
    float z = y * y, z2 = z * z;
    float even = 0.406718052498846252698f, odd = 0.403343858251329912514f;
    even = fmaf(z2, even, 0.577440339438736392009f); odd = fmaf(z2, odd, 0.961791550404184197881f);
    even = fmaf(z2, even, 2.88539009343309178325f);
    return fmaf(y, fmaf(odd, z, even), logexp);
}
__host__ __device__ inline float pow(float x, float y) { return exp2(log2(x) * y); }

}  // namespace glsl
