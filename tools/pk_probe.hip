// What do v_pk_fma_f32 and v_fma_f32 cost on gfx950?  Independent chains, one or two waves per SIMD, cycles by s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o tmp_ab/pk_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: 16 independent v_pk_fma_f32 chains; 1: 32 independent v_fma_f32 chains (the same flops); 2: v_pk_mul; 3: v_pk_add
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *cyc, int iters)
{
    v2f a[16];
    for (int i = 0; i < 16; i++) a[i] = v2f{ (float)threadIdx.x + i, 1.0f + i };
    const v2f m = { 1.0000001f, 0.9999999f }, c = { 1e-7f, -1e-7f };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(c.y));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(c.x));
                asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].y) : "v"(c.y));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out; unsigned long long *cyc;
    const int maxb = cus * 4 * 4;
    hipMalloc(&out, maxb * 64 * 4); hipMalloc(&cyc, maxb * 8);
    const int iters = 20000;
    const char *names[6] = { "16 x v_pk_fma_f32", "32 x v_fma_f32   ", "16 x v_pk_mul_f32", "16 x v_pk_add_f32", "16 x (v_mul+v_add)", "32 x v_add_f32_dpp" };
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int mode = 0; mode < 6; mode++) {
            const int blocks = cus * 4 * wps;              // wps waves on every SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
            const double instr = (double)iters * ((mode == 1 || mode >= 4) ? 32 : 16);
            const double flop = (double)blocks * 64 * iters * 16 * 2 * (mode <= 1 ? 2 : 1);
            (void)mean;
            printf("%d wave(s)/SIMD  %s: %.2f ms, %.1f TFLOP/s; %.2f ns per instruction per wave, %.2f ns of SIMD time per instruction (x 2.1 GHz = %.1f clocks)\n", wps, names[mode], ms,
                   flop / ms / 1e9, ms * 1e6 / instr, ms * 1e6 / instr / wps, ms * 1e6 / instr / wps * 2.1);
        }
    return 0;
}
