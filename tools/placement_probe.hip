// Which workgroups share a CU?  512-thread workgroups, 2 per CU (64 KiB LDS each), each records where it runs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(512) void probe(unsigned *out, unsigned long long *t)
{
    __shared__ float pad[16384];                 // 64 KiB: at most 2 workgroups per CU
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        t[blockIdx.x] = __builtin_amdgcn_s_memtime();
    }
    // hold the slot for a while so that the first 512 workgroups are co-resident
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) __builtin_amdgcn_s_sleep(20);
    if (pad[(threadIdx.x * 7) & 16383] < 0) out[0] = 0;
}
int main()
{
    const int n = 1024;
    unsigned *d; unsigned long long *dt;
    hipMalloc(&d, n * 8); hipMalloc(&dt, n * 8);
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, d, dt);
    std::vector<unsigned> h(2 * n); std::vector<unsigned long long> ht(n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> cu;
    for (int i = 0; i < 512; i++) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        unsigned cu_id = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(i);
    }
    printf("distinct CUs hosting the first 512 workgroups: %zu\n", cu.size());
    int shown = 0;
    std::map<int, int> delta;
    for (auto &kv : cu) {
        if (shown++ < 12) { printf("xcc %llu se/sh/cu %03llx:", kv.first >> 16, kv.first & 0xFFF); for (int b : kv.second) printf(" %d", b); printf("\n"); }
        if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
    }
    for (auto &kv : delta) printf("blockIdx difference %d between co-resident workgroups: %d CUs\n", kv.first, kv.second);
    return 0;
}
