// What does the SHAPE of a row-wise elementwise kernel cost on gfx950, before any arithmetic?  A 16384 x 8192 fp32 field is read
// and written once (1.07 GB: the traffic of a device copy) by kernels that differ only in how the work is handed to waves:
//   one:    a wave takes 256 columns of ONE row (16 B per lane) -- 524 288 waves, the climate step's round-5 geometry
//   rowsR:  a wave takes 256 columns of R consecutive rows, all loads first, then the stores (R = 2, 4, 8)
//   loop:   a wave streams down 256 columns of N rows, the next row's load issued before this row's store (N = 16 .. 256)
//   wide:   a wave takes 512 / 1024 columns of one row (2 / 4 x 16 B per lane)
// each also with a climate-like scalar preamble (a 64-byte record per row through the scalar cache) and with F dependent packed
// FMAs per 16 bytes as stand-in arithmetic.
// build: hipcc --offload-arch=gfx950 -O3 tools/shape_probe.hip -o tmp_ab/shape_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
struct rowrec { unsigned off[9]; int xlo, xhi, flags, L, R, p0, p1; };

template <int F>
__device__ __forceinline__ float4 work(float4 v, float s)
{
    v2f a = { v.x, v.y }, b = { v.z, v.w };
    const v2f m = { 1.0000001f, 0.9999999f }, c = { s, -s };
#pragma unroll
    for (int i = 0; i < F; i++) { a = __builtin_elementwise_fma(a, m, c); b = __builtin_elementwise_fma(b, m, c); }
    return float4{ a.x, a.y, b.x, b.y };
}

template <int R, int F, bool REC>
__global__ __launch_bounds__(256) void k_rows(const float *__restrict__ in, float *__restrict__ out, const rowrec *__restrict__ rec, int W)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * R;
    float s = 1e-7f;
    unsigned off = 0;
    if (REC) { const rowrec r = rec[y0]; off = r.off[4]; s = __int_as_float(r.flags) + 1e-7f; }
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(in) + (size_t)(off + (unsigned)(((y0 + r) * W + x0) * 4)));
#pragma unroll
    for (int r = 0; r < R; r++) *reinterpret_cast<float4 *>(out + (size_t)(y0 + r) * W + x0) = work<F>(v[r], s);
}

template <int F, bool REC>
__global__ __launch_bounds__(256) void k_loop(const float *__restrict__ in, float *__restrict__ out, const rowrec *__restrict__ rec, int W, int N)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * N;
    float s = 1e-7f;
    if (REC) { const rowrec r = rec[y0]; s = __int_as_float(r.flags) + 1e-7f; }
    const float4 *p = reinterpret_cast<const float4 *>(in + (size_t)y0 * W + x0);
    float4 *q = reinterpret_cast<float4 *>(out + (size_t)y0 * W + x0);
    const size_t pitch = (size_t)W / 4;
    float4 a = p[0], b = p[pitch];
    for (int i = 0; i < N; i += 2) {                 // two rows in flight ahead of the one being written
        const float4 c = p[(size_t)min(i + 2, N - 1) * pitch];
        q[(size_t)i * pitch] = work<F>(a, s);
        const float4 d = p[(size_t)min(i + 3, N - 1) * pitch];
        q[(size_t)(i + 1) * pitch] = work<F>(b, s);
        a = c; b = d;
    }
}

// T loads of 16 bytes per lane from the rows y-1, y, y+1 at displacements (-S, 0, S) cells (S = 0: all aligned), summed; optionally no store
template <int T, int S, bool STORE, bool BYTE>
__global__ __launch_bounds__(256) void k_taps(const float *__restrict__ in, float *__restrict__ out, const unsigned char *__restrict__ bytes, int W, int H)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    float4 v[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int yy = min(max(y + (t / 3) - 1, 0), H - 1), xx = min(max(x0 + ((t % 3) - 1) * S, 0), W - 4);
        __builtin_memcpy(&v[t], in + (size_t)yy * W + xx, 16);
    }
    float4 o = v[0];
#pragma unroll
    for (int t = 1; t < T; t++) { o.x += v[t].x; o.y += v[t].y; o.z += v[t].z; o.w += v[t].w; }
    if (BYTE) o.x += (float)bytes[((size_t)y * W + x0) >> 2];
    if (STORE || o.x == 12345.678f) *reinterpret_cast<float4 *>(out + (size_t)y * W + x0) = o;
}

// R rows per wave, the R + 2 rows it reads fetched once at three displacements; F packed FMAs per pair as stand-in arithmetic;
// a 64-byte record per row through the scalar cache first (REC)
template <int R, int S, int F, bool REC>
__global__ __launch_bounds__(256) void k_taps_rows(const float *__restrict__ in, float *__restrict__ out, const unsigned char *__restrict__ bytes,
                                                   const rowrec *__restrict__ rec, int W, int H)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y0 = blockIdx.y * R;
    float s = 1e-7f;
    unsigned off = 0;
    if (REC) { const rowrec r = rec[y0]; off = r.off[4]; s = __int_as_float(r.flags) + 1e-7f; }
    float4 v[R + 2][3];
#pragma unroll
    for (int j = 0; j < R + 2; j++)
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int yy = min(max(y0 + j - 1, 0), H - 1), xx = min(max(x0 + (t - 1) * S, 0), W - 4);
            __builtin_memcpy(&v[j][t], reinterpret_cast<const char *>(in) + (size_t)(off + (unsigned)((yy * W + xx) * 4)), 16);
        }
#pragma unroll
    for (int r = 0; r < R; r++) {
        float4 o = v[r][0];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int t = 0; t < 3; t++) if (j || t) { o.x += v[r + j][t].x; o.y += v[r + j][t].y; o.z += v[r + j][t].z; o.w += v[r + j][t].w; }
        o.x += (float)bytes[((size_t)(y0 + r) * W + x0) >> 2];
        *reinterpret_cast<float4 *>(out + (size_t)(y0 + r) * W + x0) = work<F>(o, s);
    }
}

template <int C, int F>
__global__ __launch_bounds__(256) void k_wide(const float *__restrict__ in, float *__restrict__ out, int W)
{
    const int y = blockIdx.y;
    float4 v[C];
#pragma unroll
    for (int c = 0; c < C; c++) v[c] = *reinterpret_cast<const float4 *>(in + (size_t)y * W + ((blockIdx.x * C + c) * 256 + threadIdx.x) * 4);
#pragma unroll
    for (int c = 0; c < C; c++) *reinterpret_cast<float4 *>(out + (size_t)y * W + ((blockIdx.x * C + c) * 256 + threadIdx.x) * 4) = work<F>(v[c], 1e-7f);
}

int main()
{
    const int W = 16384, H = 8192;
    const size_t n = (size_t)W * H;
    float *in, *out;
    rowrec *rec;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&rec, (size_t)H * sizeof(rowrec)));
    CK(hipMemset(in, 0, n * 4)); CK(hipMemset(out, 0, n * 4)); CK(hipMemset(rec, 0, (size_t)H * sizeof(rowrec)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; i++) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 20 < best ? ms / 20 : best;
        }
        CK(hipGetLastError());
        printf("%-44s %.3f ms  %.0f GB/s (read + write)\n", name, best, 2.0 * n * 4 / best / 1e6);
        fflush(stdout);
    };
#define TAPS(T, S, ST, B) timeit("taps T=" #T " shift=" #S " store=" #ST " byte=" #B, [&] { hipLaunchKernelGGL((k_taps<T, S, ST, B>), dim3(W / 1024, H), dim3(256), 0, 0, in, out, (const unsigned char *)rec2, W, H); })
    unsigned char *rec2;
    CK(hipMalloc(&rec2, n / 4)); CK(hipMemset(rec2, 0, n / 4));
    TAPS(1, 0, true, false); TAPS(3, 0, true, false); TAPS(9, 0, true, false); TAPS(9, 1, true, false); TAPS(9, 2, true, false); TAPS(9, 4, true, false);
    TAPS(9, 1, false, false); TAPS(9, 1, true, true); TAPS(3, 1, true, true); TAPS(5, 1, true, true); TAPS(1, 0, true, true); TAPS(1, 0, false, false);
#define TR(R, S, F, REC) timeit("taps_rows R=" #R " shift=" #S " F=" #F " rec=" #REC, [&] { hipLaunchKernelGGL((k_taps_rows<R, S, F, REC>), dim3(W / 1024, H / R), dim3(256), 0, 0, in, out, (const unsigned char *)rec2, rec, W, H); })
    TR(1, 1, 0, false); TR(1, 1, 0, true); TR(1, 1, 40, true); TR(1, 1, 80, true);
    TR(2, 1, 0, false); TR(2, 1, 0, true); TR(2, 1, 40, true); TR(2, 1, 80, true);
    TR(3, 1, 0, true); TR(3, 1, 40, true); TR(3, 1, 80, true);
    TR(4, 1, 0, true); TR(4, 1, 40, true); TR(4, 1, 80, true);
    TR(6, 1, 0, true); TR(6, 1, 40, true); TR(6, 1, 80, true);
    timeit("hipMemcpyAsync device to device", [&] { CK(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0)); });
#define ROWS(R, F, REC) timeit("rows R=" #R " F=" #F " rec=" #REC, [&] { hipLaunchKernelGGL((k_rows<R, F, REC>), dim3(W / 1024, H / R), dim3(256), 0, 0, in, out, rec, W); })
    ROWS(1, 0, false); ROWS(1, 0, true); ROWS(1, 40, true); ROWS(1, 80, true);
    ROWS(2, 0, false); ROWS(2, 0, true); ROWS(2, 40, true); ROWS(2, 80, true);
    ROWS(4, 0, false); ROWS(4, 0, true); ROWS(4, 40, true); ROWS(4, 80, true);
    ROWS(8, 0, false); ROWS(8, 0, true); ROWS(8, 40, true); ROWS(8, 80, true);
#define LOOP(N, F, REC) timeit("loop N=" #N " F=" #F " rec=" #REC, [&] { hipLaunchKernelGGL((k_loop<F, REC>), dim3(W / 1024, H / N), dim3(256), 0, 0, in, out, rec, W, N); })
    LOOP(16, 0, false); LOOP(16, 0, true); LOOP(16, 40, true); LOOP(16, 80, true);
    LOOP(32, 0, true); LOOP(32, 40, true); LOOP(32, 80, true);
    LOOP(64, 0, true); LOOP(64, 40, true); LOOP(64, 80, true);
    LOOP(256, 0, true); LOOP(256, 80, true);
#define WIDE(C, F) timeit("wide C=" #C " F=" #F, [&] { hipLaunchKernelGGL((k_wide<C, F>), dim3(W / 1024 / C, H), dim3(256), 0, 0, in, out, W); })
    WIDE(2, 0); WIDE(2, 80); WIDE(4, 0); WIDE(4, 80);
    return 0;
}
