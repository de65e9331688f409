// Developer calibration tool (not part of the product): launch overhead, clock, HBM bandwidth, MFMA rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void copy_kernel(const uint4* a, uint4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// tuned copies (the yardstick for "what a streaming kernel can reach"): UN independent 16-byte loads in flight per thread, then
// the stores; NT: non-temporal loads and stores (streamed data that is not read again)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int UN, bool NT>
__global__ __launch_bounds__(256) void copy_tuned_kernel(const uint4* __restrict__ a4, uint4* __restrict__ b4, size_t n) {
    const u32x4* a = (const u32x4*)a4; u32x4* b = (u32x4*)b4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += UN * stride) {
        u32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) if (i + u * stride < n) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) if (i + u * stride < n) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
}
template <int UN>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ a, unsigned* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += UN * stride) {
        uint4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = (i + u * stride < n) ? a[i + u * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_kernel(uint4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = make_uint4(1, 2, 3, 4);
}
__global__ void clock_kernel(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int* dp; CK(hipMalloc(&dp, 4));
    // launch overhead: N back-to-back empty kernels
    for (int rep = 0; rep < 3; ++rep) {
        const int N = 2000;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(8), dim3(256), 0, st, dp);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel x%d (8 WGs): %.2f us each (GPU events)\n", N, ms * 1e3 / N);
    }
    {   const int N = 2000;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(400), dim3(256), 65536, st, dp);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel x%d (400 WGs, 64KB LDS): %.2f us each\n", N, ms * 1e3 / N);
    }
    // clock
    unsigned long long* dc; CK(hipMalloc(&dc, 32));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(clock_kernel, dim3(1024), dim3(256), 0, st, dc, 2000000);
        CK(hipStreamSynchronize(st));
        unsigned long long h[3]; CK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
        printf("clock: memtime %llu realtime %llu -> %.0f MHz\n", h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
    }
    // bandwidth
    const size_t bytes = (size_t)1 << 30;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, st, a, b, bytes / 16);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 1 GiB: %.3f ms -> %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
    }
    // tuned copies, read-only and write-only streams (best of 3 each)
    {
        auto timeit = [&](const char* name, double moved, auto launch) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, st); launch(); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
            }
            printf("%-44s %.3f ms -> %.2f TB/s\n", name, best, moved / best / 1e9);
        };
        const size_t n = bytes / 16;
        unsigned* dout; CK(hipMalloc(&dout, 4));
        timeit("copy 1 GiB, 4 loads in flight, 2048 blocks", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tuned_kernel<4, false>), dim3(2048), dim3(256), 0, st, a, b, n); });
        timeit("copy 1 GiB, 8 loads in flight, 2048 blocks", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tuned_kernel<8, false>), dim3(2048), dim3(256), 0, st, a, b, n); });
        timeit("copy 1 GiB, 8 loads in flight, 4096 blocks", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tuned_kernel<8, false>), dim3(4096), dim3(256), 0, st, a, b, n); });
        timeit("copy 1 GiB, 8 in flight, non-temporal, 2048", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tuned_kernel<8, true>), dim3(2048), dim3(256), 0, st, a, b, n); });
        timeit("copy 128 MiB (a step's worth), 8 in flight", 2.0 * (bytes / 8), [&] { hipLaunchKernelGGL((copy_tuned_kernel<8, false>), dim3(2048), dim3(256), 0, st, a, b, n / 8); });
        timeit("read 1 GiB, 8 loads in flight", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_kernel<8>), dim3(2048), dim3(256), 0, st, a, dout, n); });
        timeit("write 1 GiB", 1.0 * bytes, [&] { hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, st, b, n); });
    }
    // MFMA
    float* mo; CK(hipMalloc(&mo, 2048 * 256 * 4));
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 20000;
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(mfma_kernel, dim3(1024), dim3(256), 0, st, mo, iters);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 1024.0 * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
        printf("mfma 16x16x32 bf16: %.3f ms -> %.1f TFLOP/s\n", ms, flop / ms / 1e9);
    }
    // host-side launch rate
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 5000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(8), dim3(256), 0, st, dp);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    printf("host launch cost: %.2f us per launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 5000);
    return 0;
}
