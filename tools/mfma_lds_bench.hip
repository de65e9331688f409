// Developer calibration: cycles per MFMA for the "A from LDS, B in registers" inner loop, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// MODE 0: MFMA only (A in regs). 1: A from LDS, prefetch P. 2: 32x32x16 MFMA only. 3: 32x32x16 with LDS
template <int MODE, int P, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void bench(unsigned long long* out, float* sink, int iters, int fill) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 28672 / 4; i += blockDim.x) ((uint32_t*)smem)[i] = fill ? (0x3f803f80u + (i * 2654435761u & 0x00ff00ffu)) : 0u;
    __syncthreads();
    uint4 b[7][2];
    for (int i = 0; i < 7; ++i) for (int g = 0; g < 2; ++g) { uint32_t v = fill ? 0x3f803f80u + ((lane * 7 + i * 13 + g) & 0xff) : 0u; b[i][g] = make_uint4(v, v + 1, v + 2, v + 3); }
    f32x4 acc[4][2];
    for (int t = 0; t < 4; ++t) for (int g = 0; g < 2; ++g) acc[t][g] = (f32x4){0, 0, 0, 0};
    f32x16 acc32[2];
    for (int g = 0; g < 2; ++g) for (int i = 0; i < 16; ++i) acc32[g][i] = 0;
    const char* lb = smem + (lane & 15) * 64 + (lane >> 4) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 28; ++i)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[i & 3][g] = mfma16(b[(i + 1) % 7][g ^ 1], b[i >> 2][g], acc[i & 3][g]);
        } else if (MODE == 1) {
            uint4 av[P];
#pragma unroll
            for (int i = 0; i < P; ++i) av[i] = *(const uint4*)(lb + i * 1024);
#pragma unroll
            for (int i = 0; i < 28; ++i) {
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[i & 3][g] = mfma16(av[i % P], b[i >> 2][g], acc[i & 3][g]);
                if (i + P < 28) av[i % P] = *(const uint4*)(lb + (i + P) * 1024);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 14; ++i)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    acc32[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b[(i + 1) % 7][g ^ 1]), __builtin_bit_cast(bf16x8_t, b[i % 7][g]), acc32[g], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int g = 0; g < 2; ++g) s += acc[t][g][0];
    for (int g = 0; g < 2; ++g) s += acc32[g][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE, int P, int W>
int run(const char* name, int blocks, int fill) {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 8)); CK(hipMalloc(&sink, (size_t)blocks * 256 * W * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((bench<MODE, P, W>), dim3(blocks), dim3(256 * W), 98304, 0, d, sink, iters, fill);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        const double nm = (MODE == 2) ? 28.0 : 56.0;
        const double flop = (double)blocks * 4 * W * iters * nm * (MODE == 2 ? 32768.0 : 16384.0);
        if (rep) printf("%-34s blocks %4d fill %d: %.1f memtime-cycles/MFMA, %.2f ms, %.0f TF, clock %.0f MHz\n", name, blocks, fill, (double)h / (iters * nm), ms,
                        flop / ms / 1e9, (double)h / (ms * 1e3));
    }
    return 0;
}
int main() {
    for (int fill = 0; fill < 2; ++fill) {
        run<0, 4, 1>("16x16x32 regs, 1 wave/SIMD", 256, fill);
        run<0, 4, 2>("16x16x32 regs, 2 waves/SIMD", 256, fill);
        run<1, 4, 1>("16x16x32 LDS P=4, 1 wave/SIMD", 256, fill);
        run<1, 8, 1>("16x16x32 LDS P=8, 1 wave/SIMD", 256, fill);
        run<1, 8, 2>("16x16x32 LDS P=8, 2 waves/SIMD", 256, fill);
        run<2, 4, 1>("32x32x16 regs, 1 wave/SIMD", 256, fill);
        run<0, 4, 1>("16x16x32 regs, 1 wave/SIMD, 1 blk", 1, fill);
        run<1, 8, 1>("16x16x32 LDS P=8, 1 wave, 1 blk", 1, fill);
    }
    return 0;
}
