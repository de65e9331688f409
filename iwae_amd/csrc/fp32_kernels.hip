// float32 mode (iwae_config.precision = IWAE_PREC_FP32): the reference's arithmetic -- Keras Dense layers in float32
// (src/iwae1.py:31-34,72-75) -- with every GEMM on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit a k-ordered
// fmaf chain, no reduced-precision operands).  Row-major float32 activations in their natural widths, weights read straight
// from the fp32 master parameters in Keras [in, out] order.  This mode exists for parity (scalars rel 1e-5,
// gradients rel 1e-4 against the float64 oracle) and for the k = 5000 evaluator; the bf16 kernels in kernels.hip are the fast path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <algorithm>
#include "kernels.h"

namespace iwae {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

// tanh through one hardware exp2: 1 - 2/(1 + e^{2|x|}), sign restored.  Absolute error <= ~1.5e-7 (one ulp of 1.0; saturates cleanly: e^{2|x|} = inf
// gives exactly 1) -- the library tanhf is ~50 instructions per value, 20 million values per tanh layer of the full-size step.
__device__ __forceinline__ float tanh_f32(float x) {
    const float t = __expf(2.0f * fabsf(x));
    return __builtin_copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f), x);
}

// a fetched quad of op(B) times the row weights of its k index / indices (GemmF32Args.brow_scale)
__device__ __forceinline__ float4 scale_b_quad(const GemmF32Args& a, float4 t, int gk, bool b_nfast) {
    if (b_nfast) { const float w = a.brow_scale[gk]; return make_float4(t.x * w, t.y * w, t.z * w, t.w * w); }
    return make_float4(t.x * a.brow_scale[gk], t.y * a.brow_scale[gk + 1], t.z * a.brow_scale[gk + 2], t.w * a.brow_scale[gk + 3]);
}

// C[M,N] (=|+=) epi(op(A)[M,K] op(B)[K,N] + bias): 64 x 64 tile per 256-thread workgroup, K walked in steps of 16 through
// LDS; wave (wm, wn) owns a 32 x 32 sub-tile = 2 x 2 MFMA tiles.  Element (m,k) of A is A[m*sam + k*sak], (k,n) of B is
// B[k*sbk + n*sbn]: plain, transposed-A (weight gradient X^T G) and transposed-B (dX = G W^T) products are the same kernel;
// the tile loads walk the unit-stride index fastest so they stay coalesced either way.
// blockIdx.z = K split: split z covers k in [z*kchunk, (z+1)*kchunk) and writes C + z*slab_stride (fp32 slabs, summed in a
// fixed order by reduce_slabs_f32_kernel: deterministic, no float atomics).
// KB = k-steps of 16 per loop iteration (1 or 2): with few workgroups (the encoder's layers on the batch's 1 024 images: 64 of them, one per CU) an iteration is
// one exposed round trip -- fetch, stash, barrier, 16 MFMAs -- and K = 784 is 49 of them (22 us per launch); KB = 2 makes it 25 (round 3).
template <int KB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args a) {
    constexpr int BK = 16 * KB;
    __shared__ float sA[64][BK + 1];
    __shared__ float sB[BK][80];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_beg = blockIdx.z * a.kchunk, k_end = min(a.K, k_beg + a.kchunk);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    f32x4v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    // Each thread fetches KB quads (4 consecutive elements along the unit-stride index) of the A tile and of the B tile per iteration, as
    // float4 where the host found the operand 16-byte aligned (a.avec / a.bvec), and the NEXT iteration's quads are requested before
    // this one's MFMAs: the global latency hides under them.
    const bool a_kfast = a.sak == 1, b_nfast = a.sbn == 1;
    auto a_pos = [&](int u, int& lm, int& lk) { const int qq = tid + 256 * u; if (a_kfast) { lm = qq / (4 * KB); lk = (qq % (4 * KB)) * 4; } else { lk = qq >> 4; lm = (qq & 15) * 4; } };
    auto b_pos = [&](int u, int& ln, int& lk) { const int qq = tid + 256 * u; if (b_nfast) { lk = qq >> 4; ln = (qq & 15) * 4; } else { ln = qq / (4 * KB); lk = (qq % (4 * KB)) * 4; } };
    auto fetch_a = [&](int k0, int u) -> float4 {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int lm, lk;
        a_pos(u, lm, lk);
        const int gm = m0 + lm, gk = k0 + lk;
        const float* p = a.A + (size_t)gm * a.sam + (size_t)gk * a.sak;
        const int lim = a_kfast ? k_end - gk : a.M - gm;             // elements of the quad that exist
        const bool outer_ok = a_kfast ? gm < a.M : gk < k_end;
        if (outer_ok && lim >= 4 && a.avec) { const float4 t = *(const float4*)p; return t; }
        if (outer_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < lim) v[e] = p[e];
        }
        if (a.Cones) {      // the row of ones behind the last row of op(A)
            if (a_kfast) { if (gm == a.M) { for (int e = 0; e < 4; ++e) v[e] = (gk + e < k_end) ? 1.0f : 0.0f; } }
            else if (gk < k_end) { for (int e = 0; e < 4; ++e) if (gm + e == a.M) v[e] = 1.0f; }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto fetch_b = [&](int k0, int u) -> float4 {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int ln, lk;
        b_pos(u, ln, lk);
        const int gn = n0 + ln, gk = k0 + lk;
        const float* p = a.B + (size_t)gk * a.sbk + (size_t)gn * a.sbn;
        const int lim = b_nfast ? a.N - gn : k_end - gk;
        const bool outer_ok = b_nfast ? gk < k_end : gn < a.N;
        if (outer_ok && lim >= 4 && a.bvec) { float4 t = *(const float4*)p; if (a.brow_scale) t = scale_b_quad(a, t, gk, b_nfast); return t; }
        if (outer_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < lim) v[e] = p[e] * (a.brow_scale ? a.brow_scale[b_nfast ? gk : gk + e] : 1.0f);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    float4 ra[KB], rb[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) { ra[u] = make_float4(0.f, 0.f, 0.f, 0.f); rb[u] = ra[u]; }
    if (k_beg < k_end) {
#pragma unroll
        for (int u = 0; u < KB; ++u) { ra[u] = fetch_a(k_beg, u); rb[u] = fetch_b(k_beg, u); }
    }
    for (int k0 = k_beg; k0 < k_end; k0 += BK) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const float av4[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w}, bv4[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
            int lm, lka, ln, lkb;
            a_pos(u, lm, lka);
            b_pos(u, ln, lkb);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a_kfast) sA[lm][lka + e] = av4[e]; else sA[lm + e][lka] = av4[e];
                if (b_nfast) sB[lkb][ln + e] = bv4[e]; else sB[lkb + e][ln] = bv4[e];
            }
        }
        __syncthreads();
        if (k0 + BK < k_end) {
#pragma unroll
            for (int u = 0; u < KB; ++u) { ra[u] = fetch_a(k0 + BK, u); rb[u] = fetch_b(k0 + BK, u); }
        }
#pragma unroll
        for (int kk = 0; kk < 4 * KB; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = sA[32 * wm + 16 * i + r16][4 * kk + q];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = sB[4 * kk + q][32 * wn + 16 * j + r16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* C = a.C + (size_t)blockIdx.z * a.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * wn + 16 * j + r16;
            if (n >= a.N) continue;
            const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wm + 16 * i + 4 * q + r;
                if (m >= a.M) {
                    if (a.Cones && m == a.M) a.Cones[(size_t)blockIdx.z * a.cones_stride + n] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] + bias;
                if (a.orow_scale) v *= a.orow_scale[m];
                if (a.epi == GEMM_EPI_TANH) v = tanh_f32(v);                    // iwae1.py:31-32,72-73
                else if (a.epi == GEMM_EPI_EXP) v = expf(v) + 1e-6f;             // iwae1.py:34,42
                else if (a.epi == GEMM_EPI_DTANH) { const float y = a.ACT[(size_t)m * a.ldact + n]; v *= 1.0f - y * y; }
                float* dst = C + (size_t)m * a.ldc + n;
                *dst = a.accumulate ? *dst + v : v;
            }
        }
}

// The same product on 128 x 128 tiles (round 3), for every GEMM that has more than one 64-tile in both directions: wave (wm, wn)
// owns 64 x 64 = 4 x 4 MFMA tiles, so a k-step of 4 costs it 8 LDS reads for 16 MFMAs (the 64 x 64 kernel: 4 for 4 -- its MFMA pipe
// was 39 % busy) and the activations of the 784-wide layer are fetched 7 times instead of 13; the k-steps of 16 are double
// buffered in LDS (one barrier per step), the next step's quads are in flight under this step's 64 MFMAs.  Same operand addressing
// (element strides), same K split, same epilogues; bit-for-bit the same k-ordered fmaf chain per element.
// TM x TN = MFMA tiles per wave (2 x 2 waves): <4, 4> is the 128 x 128 tile; <2, 7> = 64 x 224 and <7, 2> = 224 x 64 take the products in which a
// 200-wide dimension would fill a 128-tile pair to 78 % (round 3: most products of the step).
template <int TM, int TN, bool a_kfast, bool b_nfast>      // (a_kfast = op(A)'s k index is the unit-stride one, b_nfast = op(B)'s n index: compile-time, the position arithmetic folds)
__global__ __launch_bounds__(256, 4) void gemm_f32_big_kernel(GemmF32Args a) {      // <= 128 registers: four workgroups per CU (1 024 slots: the 800 workgroups of a 51 200 x 200 product are one round, not two)
    constexpr int BM = 32 * TM, BN = 32 * TN, BNP = BN + 16;      // (BNP = 16 mod 32: a half wave's B reads of k and k + 1 fall into different banks)
    constexpr int NA = (BM * 4 + 255) / 256, NB = (BN * 4 + 255) / 256, QM = BM / 4, QN = BN / 4;      // quads per thread and operand; quads per k row (m- / n-fast)
    __shared__ float sA[2][BM][17];
    __shared__ float sB[2][16][BNP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_beg = blockIdx.z * a.kchunk, k_end = min(a.K, k_beg + a.kchunk);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    f32x4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    // quad u of a thread: number qq = tid + 256 u of the tile's quads; k-fast: 4 quads per row (m = qq >> 2, k = 4 (qq & 3)); m-fast: QM quads per k
    // (k = qq / QM, m = 4 (qq % QM)); the B tile likewise
    auto a_pos = [&](int u, int& lm, int& lk) { const int qq = tid + 256 * u; if (a_kfast) { lm = qq >> 2; lk = (qq & 3) * 4; } else { lk = qq / QM; lm = (qq % QM) * 4; } return qq < BM * 4; };
    auto b_pos = [&](int u, int& ln, int& lk) { const int qq = tid + 256 * u; if (b_nfast) { lk = qq / QN; ln = (qq % QN) * 4; } else { ln = qq >> 2; lk = (qq & 3) * 4; } return qq < BN * 4; };
    auto fetch_a = [&](int k0, int u) -> float4 {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int lm, lk;
        if (!a_pos(u, lm, lk)) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int gm = m0 + lm, gk = k0 + lk;
        const float* p = a.A + (size_t)gm * a.sam + (size_t)gk * a.sak;
        const int lim = a_kfast ? k_end - gk : a.M - gm;
        const bool outer_ok = a_kfast ? gm < a.M : gk < k_end;
        if (outer_ok && lim >= 4 && a.avec) { const float4 t = *(const float4*)p; return t; }
        if (outer_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < lim) v[e] = p[e * (a_kfast ? a.sak : a.sam)];
        }
        if (a.Cones) {      // the row of ones behind the last row of op(A)
            if (a_kfast) { if (gm == a.M) { for (int e = 0; e < 4; ++e) v[e] = (gk + e < k_end) ? 1.0f : 0.0f; } }
            else if (gk < k_end) { for (int e = 0; e < 4; ++e) if (gm + e == a.M) v[e] = 1.0f; }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto fetch_b = [&](int k0, int u) -> float4 {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int ln, lk;
        if (!b_pos(u, ln, lk)) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int gn = n0 + ln, gk = k0 + lk;
        const float* p = a.B + (size_t)gk * a.sbk + (size_t)gn * a.sbn;
        const int lim = b_nfast ? a.N - gn : k_end - gk;
        const bool outer_ok = b_nfast ? gk < k_end : gn < a.N;
        if (outer_ok && lim >= 4 && a.bvec) { float4 t = *(const float4*)p; if (a.brow_scale) t = scale_b_quad(a, t, gk, b_nfast); return t; }
        if (outer_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < lim) v[e] = p[e * (b_nfast ? a.sbn : a.sbk)] * (a.brow_scale ? a.brow_scale[b_nfast ? gk : gk + e] : 1.0f);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto stash = [&](int buf, const float4 (&ra)[NA], const float4 (&rb)[NB]) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            int lm, lk;
            if (!a_pos(u, lm, lk)) continue;
            const float av4[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (a_kfast) sA[buf][lm][lk + e] = av4[e]; else sA[buf][lm + e][lk] = av4[e]; }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            int ln, lk;
            if (!b_pos(u, ln, lk)) continue;
            const float bv4[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (b_nfast) sB[buf][lk][ln + e] = bv4[e]; else sB[buf][lk + e][ln] = bv4[e]; }
        }
    };
    float4 ra[NA], rb[NB];
#pragma unroll
    for (int u = 0; u < NA; ++u) ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NB; ++u) rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k_beg < k_end) {
#pragma unroll
        for (int u = 0; u < NA; ++u) ra[u] = fetch_a(k_beg, u);
#pragma unroll
        for (int u = 0; u < NB; ++u) rb[u] = fetch_b(k_beg, u);
    }
    stash(0, ra, rb);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_beg; k0 < k_end; k0 += 16) {
        const bool more = k0 + 16 < k_end;
        if (more) {
#pragma unroll
            for (int u = 0; u < NA; ++u) ra[u] = fetch_a(k0 + 16, u);
#pragma unroll
            for (int u = 0; u < NB; ++u) rb[u] = fetch_b(k0 + 16, u);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = sA[buf][16 * TM * wm + 16 * i + r16][4 * kk + q];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = sB[buf][4 * kk + q][16 * TN * wn + 16 * j + r16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1, ra, rb);
        __syncthreads();
        buf ^= 1;
    }
    if constexpr (TM == 4 && TN == 4) {
    if (a.epi == GEMM_EPI_BERN) {       // the output layer: log p(x|z) of this half tile's 64 columns per row, no logits in HBM
        float bias4[4];
        bool nok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 64 * wn + 16 * j + r16;
            nok[j] = n < a.N;
            bias4[j] = (nok[j] && a.bias) ? a.bias[n] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 64 * wm + 16 * i + 4 * q + r;
                const int mc = min(m, a.M - 1);
                const float* xr = a.XB + (size_t)(mc / a.bern_k) * a.bern_X;
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (nok[j]) {
                        const float l = acc[i][j][r] + bias4[j];
                        const float xv = xr[n0 + 64 * wn + 16 * j + r16];
                        // e = exp(-|l|) once, for softplus(l) = max(l,0) + log(1 + e) and sigmoid(l) = (l >= 0 ? 1 : e) / (1 + e).  Hardware exp2 / log2
                        // (1 ulp each): the absolute error of a term is <= 1e-7 (log(1 + e) loses RELATIVE accuracy only where e < 1e-7, i.e. where the
                        // term itself is < 1e-7) against sums of O(100) -- the float32 parity tolerances (scalars 1e-5, gradients 1e-4 relative) are far above
                        const float e = __expf(-fabsf(l)), ope = 1.0f + e;
                        sum += xv * l - (fmaxf(l, 0.0f) + __logf(ope));      // iwae1.py:111
                        // training step: s = x - sigmoid(l) stays where the logits would have gone
                        if (a.C && m < a.M) a.C[(size_t)m * a.ldc + n0 + 64 * wn + 16 * j + r16] = xv - (l >= 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(ope);
                    }
                }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);      // the 16 columns of a tile sit on lanes r16
                if (r16 == 0 && m < a.M) a.part[(size_t)(2 * blockIdx.x + wn) * a.part_stride + m] = sum;
            }
        return;
    }
    }
    float* C = a.C + (size_t)blockIdx.z * a.slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + 16 * TN * wn + 16 * j + r16;
            if (n >= a.N) continue;
            const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * TM * wm + 16 * i + 4 * q + r;
                if (m >= a.M) {
                    if (a.Cones && m == a.M) a.Cones[(size_t)blockIdx.z * a.cones_stride + n] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] + bias;
                if (a.orow_scale) v *= a.orow_scale[m];
                if (a.epi == GEMM_EPI_TANH) v = tanh_f32(v);
                else if (a.epi == GEMM_EPI_EXP) v = expf(v) + 1e-6f;
                else if (a.epi == GEMM_EPI_DTANH) { const float y = a.ACT[(size_t)m * a.ldact + n]; v *= 1.0f - y * y; }
                float* dst = C + (size_t)m * a.ldc + n;
                *dst = a.accumulate ? *dst + v : v;
            }
        }
}

// ---------------------------------------------------------------------------------
// gemm_f32_v2_kernel (round 5): the same products, tiles, K split and epilogues as gemm_f32_big_kernel with a k loop that leaves the vector
// issue to the MFMAs.  Counters of the old loop (profiles/r05_f32_gemm_counters.txt): 4.45 vector instructions per MFMA -- 249 per wave and
// k-step of 16 against 56 MFMAs: the operands' addresses, bounds and alignment re-derived per fetched quad and k-step (64-bit multiplies), a scalar
// stash (four ds_write_b32 per quad, 53 % of the LDS cycles bank conflicts) -- and the matrix pipe 50 % busy.  Here:
//   * a quad's global pointer is made once and advanced by a constant per k-step; whether it is inside the matrix, aligned (one 16-byte load), outside
//     (zeros) or on an edge (element by element, with every bound -- also the whole last k-step of a ragged K) is a per-thread state made once;
//   * one ds_write_b128 per quad, conflict-free: an operand whose k index is the unit-stride one sits in LDS as [row][16 k] with the four 16-byte
//     slots of a row XOR-ed by swz(row) -- so that a lane reads its four k values of the k-step as ONE ds_read_b128 without bank conflicts in that
//     instruction's 16-lane groups (MI355X_MICROARCH.md, LDS table) -- the other kind as [16 k][extent + 4] (pitch = 4 mod 8: the two quads of a
//     ds_read_b32 half fall into different halves of the banks);
//   * lane quad q contracts k = 4 q + j in MFMA step j of a k-step (any assignment of the 16 k's to (q, j) is a valid product; this one is what the
//     16-byte reads deliver) -- float32 results differ from gemm_f32_big_kernel's in the order of the sum only;
//   * the operand with fewer MFMA tiles per wave is held in registers for the k-step, the other streams through two-tile groups read one group ahead.
// ---------------------------------------------------------------------------------
#ifdef IWAE_DENSE_STAMPS      // diagnostic build (STAMPS=1): per-wave cycle sums of the kernel's phases -> a.stamps[wave][8]
#define GS_STAMP(slot)                                                                 \
    {                                                                                  \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        gs_sum[slot] += t_ - gs_prev;                                                  \
        gs_prev = t_;                                                                  \
    }
#else
#define GS_STAMP(slot)
#endif
__device__ __forceinline__ int gemm_swz16(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }      // (0, 2, 3, 1) for rows 0-3, 4-7, 8-11, 12-15 of a 16-row tile
// a quad off the fast path: kfast: elements (e, k .. k + 3), else (e .. e + 3, k); zero outside the matrix / beyond k_end; `ones`: row e == Ext is all ones
__device__ __forceinline__ float4 gemm_f32_slow_quad(const float* base, long s_e, long s_k, int e, int k, int Ext, int k_end, bool kfast, bool ones) {
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ee = kfast ? e : e + t, kk = kfast ? k + t : k;
        float x = 0.0f;
        if (kk < k_end) {
            if (ee < Ext) x = base[(size_t)ee * s_e + (size_t)kk * s_k];
            else if (ones && ee == Ext) x = 1.0f;
        }
        v[t] = x;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
// An accumulator tile holds, per lane (n16, q), rows 4q .. 4q + 3 of column n16.  Transposed inside every group of four lanes (two exchanges with
// lane ^ 1 and lane ^ 2), lane (n16 = 4c + p, q) holds row 4q + p, columns 4c .. 4c + 3: ONE 16-byte store (and one 16-byte load per epilogue
// operand) per tile instead of four dword ones -- what a vector-memory instruction costs a CU is the instruction.
__device__ __forceinline__ void quad_transpose(float (&x)[4], int lane) {
    const int p = lane & 3;
    float t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[r ^ 1]), 0xB1, 0xF, 0xF, true));      // lane ^ 1
        t[r] = ((r ^ p) & 1) ? o : x[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t[r ^ 2]), 0x4E, 0xF, 0xF, true));      // lane ^ 2
        x[r] = ((r ^ p) & 2) ? o : t[r];
    }
}
// WGM x WGN = the workgroup's waves (2 x 2: the tiles of gemm_f32_big_kernel; 4 x 2 / 2 x 4: 8 waves on a 128 x 224 / 224 x 128 tile).  The k loop is
// bound by what a CU can request per clock (ablations, profiles/r05_f32_gemm_ablations.txt: the requests alone take as long as the MFMAs alone,
// and the two overlap badly): a tile twice as tall moves 39 % fewer operand bytes per FLOP.
template <int TM, int TN, int WGM, int WGN, bool AK, bool BNF, int OCC = 4>      // AK: op(A)'s k index is the unit-stride one; BNF: op(B)'s n index is; OCC: waves per SIMD the registers allow
__global__ __launch_bounds__(64 * WGM * WGN, OCC) void gemm_f32_v2_kernel(GemmF32Args a) {
    constexpr int NT = 64 * WGM * WGN;
    [[maybe_unused]] constexpr int NW = WGM * WGN;
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int PA = AK ? 16 : BM + 4, PB = BNF ? BN + 4 : 16;                    // row pitch in LDS (floats)
    constexpr int SA = AK ? BM * 16 : 16 * PA, SB = BNF ? 16 * PB : BN * 16;        // floats per buffer
    constexpr int QM = BM / 4, QN = BN / 4;
    constexpr bool HOLD_A = TM <= TN;
    __shared__ __attribute__((aligned(16))) float sA[2][SA];
    __shared__ __attribute__((aligned(16))) float sB[2][SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef IWAE_DENSE_STAMPS
    unsigned long long gs_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gs_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gs_prev)::"memory");
#endif
    // K-split launches: the output tiles of ONE split read the same rows of both operands -- workgroups go to the 8 XCDs round-robin by their linear
    // index, so the index is re-read such that a split's tiles are neighbours on one XCD (its L2 then serves the re-reads: counters of the output
    // layer's weight gradient had 505 MB fetched for 201 MB of operands, 13 column tiles of a split on 8 different L2s)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int nb = gridDim.x * gridDim.y * gridDim.z;
        if (gridDim.z > 1 && (nb & 7) == 0) {
            const int L = bx + gridDim.x * (by + gridDim.y * bz), V = (L & 7) * (nb >> 3) + (L >> 3);
            bx = V % gridDim.x; by = (V / gridDim.x) % gridDim.y; bz = V / (gridDim.x * gridDim.y);
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int k_beg = bz * a.kchunk, k_end = min(a.K, k_beg + a.kchunk);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r16 = lane & 15, q = lane >> 4;
    f32x4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    // ---- the thread's quads.  Quad u of an operand tile [E = BM | BN][16 k]:
    //   k index unit-stride: row e = (tid >> 2) + 64 u, k = 4 (tid & 3) .. + 3            (E / 64 quads per thread, rounded up)
    //   otherwise:          k row = (256 / TPR) u + tid / TPR, e = 4 (tid % TPR) .. + 3    (TPR = threads per k row: the power of two >= E / 4)
    // -- either way quad u + 1 is a CONSTANT away from quad u, in memory and in LDS: one pointer, one LDS offset and two state bits per quad
    // (0 outside: zeros, 1 one 16-byte load, 2 element by element, 3 no such quad) are all a thread keeps per operand.
    constexpr int TPRA = QM > 32 ? 64 : QM > 16 ? 32 : 16, TPRB = QN > 32 ? 64 : QN > 16 ? 32 : 16;      // (with NT threads: NT / 4 rows, NT / TPR k rows per quad index)
    constexpr int RQ = NT / 4, KRA = NT / TPRA, KRB = NT / TPRB;
    constexpr int NA = AK ? (BM + RQ - 1) / RQ : (16 + KRA - 1) / KRA, NB = BNF ? (16 + KRB - 1) / KRB : (BN + RQ - 1) / RQ;
    const bool ones = a.Cones != nullptr;
    const int eA0 = AK ? tid >> 2 : 4 * (tid % TPRA), kA0 = AK ? 4 * (tid & 3) : tid / TPRA;
    const int eB0 = BNF ? 4 * (tid % TPRB) : tid >> 2, kB0 = BNF ? tid / TPRB : 4 * (tid & 3);
    constexpr int dEA = AK ? RQ : 0, dKA = AK ? 0 : KRA, dEB = BNF ? 0 : RQ, dKB = BNF ? KRB : 0;      // quad u -> u + 1
    const int oA0 = AK ? eA0 * 16 + (((tid & 3) ^ gemm_swz16(eA0)) << 2) : kA0 * PA + eA0;
    const int oB0 = BNF ? kB0 * PB + eB0 : eB0 * 16 + (((tid & 3) ^ gemm_swz16(eB0)) << 2);
    constexpr int dOA = AK ? RQ * 16 : dKA * PA, dOB = BNF ? dKB * PB : RQ * 16;
    const float* pA = a.A + (size_t)(m0 + eA0) * a.sam + (size_t)(k_beg + kA0) * a.sak;
    const float* pB = a.B + (size_t)(n0 + eB0) * a.sbn + (size_t)(k_beg + kB0) * a.sbk;
    const long dPA = AK ? RQ * a.sam : (long)dKA * a.sak, dPB = BNF ? (long)dKB * a.sbk : RQ * a.sbn;
    unsigned stA = 0, stB = 0;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int le = eA0 + dEA * u, ge = m0 + le;
        const bool exists = AK ? le < BM : le < BM && kA0 + dKA * u < 16;
        const bool inside = AK ? ge < a.M : ge + 3 < a.M, outside = ge >= a.M + (ones ? 1 : 0);
        stA |= (unsigned)(!exists ? 3 : (inside && a.avec) ? 1 : outside ? 0 : 2) << (2 * u);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int le = eB0 + dEB * u, ge = n0 + le;
        const bool exists = BNF ? le < BN && kB0 + dKB * u < 16 : le < BN;
        const bool inside = BNF ? ge + 3 < a.N : ge < a.N, outside = ge >= a.N;
        stB |= (unsigned)(!exists ? 3 : (inside && a.bvec) ? 1 : outside ? 0 : 2) << (2 * u);
    }
    const long stepA = 16 * a.sak, stepB = 16 * a.sbk;
    float4 ra[NA], rb[NB], rw[NB];
    // the quads of the k-step at k0 (full: all 16 k's exist); the pointers move on to the next k-step
    auto fetch = [&](int k0, bool full) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const unsigned st = (stA >> (2 * u)) & 3u;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st == 1 && full) t = *(const float4*)(pA + u * dPA);
            else if (st == 2 || (st == 1 && !full)) t = gemm_f32_slow_quad(a.A, a.sam, a.sak, m0 + eA0 + dEA * u, k0 + kA0 + dKA * u, a.M, k_end, AK, ones);
            ra[u] = t;
        }
        pA += stepA;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const unsigned st = (stB >> (2 * u)) & 3u;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st == 1 && full) t = *(const float4*)(pB + u * dPB);
            else if (st == 2 || (st == 1 && !full)) t = gemm_f32_slow_quad(a.B, a.sbn, a.sbk, n0 + eB0 + dEB * u, k0 + kB0 + dKB * u, a.N, k_end, !BNF, false);
            rb[u] = t;
            // row k of op(B) counts with weight brow_scale[k] (wave-uniform branch): the weights are REQUESTED here and multiplied in when the quads go
            // to LDS -- multiplied here, the wait for a weight was a wait for every quad requested before it (vector-memory results return in order):
            // the whole fetch latency sat in front of the k-step's MFMAs (phase stamps: 64 % of the output layer's weight gradient)
            if (a.brow_scale && st != 3) {
                const int gk = k0 + kB0 + dKB * u;
                if (BNF) { const float w = gk < k_end ? a.brow_scale[gk] : 0.0f; rw[u] = make_float4(w, w, w, w); }
                else rw[u] = make_float4(gk < k_end ? a.brow_scale[gk] : 0.0f, gk + 1 < k_end ? a.brow_scale[gk + 1] : 0.0f,
                                         gk + 2 < k_end ? a.brow_scale[gk + 2] : 0.0f, gk + 3 < k_end ? a.brow_scale[gk + 3] : 0.0f);
            }
        }
        pB += stepB;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NA; ++u) if (((stA >> (2 * u)) & 3u) != 3u) *(float4*)&sA[buf][oA0 + dOA * u] = ra[u];
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (((stB >> (2 * u)) & 3u) != 3u) {
                float4 t = rb[u];
                if (a.brow_scale) { t.x *= rw[u].x; t.y *= rw[u].y; t.z *= rw[u].z; t.w *= rw[u].w; }
                *(float4*)&sB[buf][oB0 + dOB * u] = t;
            }
    };
    // ---- fragment addresses of the lane: tile t of the wave's TM (TN) tiles, step j
    const int arow = 16 * TM * wm + r16, brow = 16 * TN * wn + r16;
    const int afr = AK ? arow * 16 + ((q ^ gemm_swz16(r16)) << 2) : 4 * q * PA + arow;      // (+ 16 * 16 t | + 16 t, + j * PA)
    const int bfr = BNF ? 4 * q * PB + brow : brow * 16 + ((q ^ gemm_swz16(r16)) << 2);
    auto frag_a = [&](const float* As, int t, float (&v)[4]) {
        if (AK) { const float4 x = *(const float4*)(As + afr + 256 * t); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = As[afr + 16 * t + j * PA];
        }
    };
    auto frag_b = [&](const float* Bs, int t, float (&v)[4]) {
        if (BNF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = Bs[bfr + 16 * t + j * PB];
        } else { const float4 x = *(const float4*)(Bs + bfr + 256 * t); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
    };
    auto compute = [&](int buf) {
        const float *As = sA[buf], *Bs = sB[buf];
        if constexpr (HOLD_A) {
            float ah[TM][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) frag_a(As, i, ah[i]);
            constexpr int NG = (TN + 1) / 2;
            float cur[2][4], nxt[2][4];
            frag_b(Bs, 0, cur[0]);
            if (TN > 1) frag_b(Bs, 1, cur[1]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) { frag_b(Bs, 2 * g + 2, nxt[0]); if (2 * g + 3 < TN) frag_b(Bs, 2 * g + 3, nxt[1]); }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (2 * g + h < TN) {
#pragma unroll
                            for (int i = 0; i < TM; ++i) acc[i][2 * g + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][j], cur[h][j], acc[i][2 * g + h], 0, 0, 0);
                        }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cur[h][j] = nxt[h][j];
            }
        } else {
            float bh[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j) frag_b(Bs, j, bh[j]);
            constexpr int NG = (TM + 1) / 2;
            float cur[2][4], nxt[2][4];
            frag_a(As, 0, cur[0]);
            if (TM > 1) frag_a(As, 1, cur[1]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) { frag_a(As, 2 * g + 2, nxt[0]); if (2 * g + 3 < TM) frag_a(As, 2 * g + 3, nxt[1]); }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (2 * g + h < TM) {
#pragma unroll
                            for (int n = 0; n < TN; ++n) acc[2 * g + h][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[h][j], bh[n][j], acc[2 * g + h][n], 0, 0, 0);
                        }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cur[h][j] = nxt[h][j];
            }
        }
    };
#pragma unroll
    for (int u = 0; u < NA; ++u) ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NB; ++u) rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k_beg < k_end) fetch(k_beg, k_beg + 16 <= k_end);
    stash(0);
    __syncthreads();
    int buf = 0;
#ifdef IWAE_DIAG
#define GEMM_DBG(bit) (a.dbg & (bit))
#else
#define GEMM_DBG(bit) false
#endif
    GS_STAMP(0)      // set-up, first k-step's quads in LDS
    for (int k0 = k_beg; k0 < k_end; k0 += 16) {
        const bool more = k0 + 16 < k_end;
        if (more && !GEMM_DBG(1)) fetch(k0 + 16, k0 + 32 <= k_end);
        GS_STAMP(1)      // requests of the next k-step
        if (!GEMM_DBG(4)) compute(buf);
        GS_STAMP(2)      // fragment reads + MFMAs issued
        if (more && !GEMM_DBG(2)) {
#ifdef IWAE_DENSE_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GS_STAMP(3)  // wait for the requested quads
#endif
            stash(buf ^ 1);
        }
        GS_STAMP(4)      // LDS stores
        if (!GEMM_DBG(16)) __syncthreads();
        GS_STAMP(5)      // barrier
        if (!GEMM_DBG(8)) buf ^= 1;
    }
    if constexpr (TM == 4 && TN == 4 && WGM == 2 && WGN == 2) {
    if (a.epi == GEMM_EPI_BERN) {       // the output layer: log p(x|z) of this half tile's 64 columns per row, no logits in HBM (see gemm_f32_big_kernel)
        // Transposed tiles (quad_transpose): a lane holds row 4q + p, columns 4c .. 4c + 3 of each of the wave's 4 x 4 tiles -- x, the bias and s = x - sigmoid(l)
        // go as 16-byte accesses, the 16 logarithms of a lane's row as ONE log of the product of (1 + e^-|l|) <= 2^16, and the row's 64 columns
        // meet across the four lanes c with two shuffles.
        if (a.cvec && (a.bern_X & 3) == 0 && (((uintptr_t)a.XB) & 15) == 0) {
            const int p = lane & 3, cq = 4 * (r16 >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 64 * wm + 16 * i + 4 * q + p;
                const int mc = min(m, a.M - 1);
                const float* xr = a.XB + (size_t)(mc / a.bern_k) * a.bern_X;
                float sum = 0.0f, prod = 1.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + 64 * wn + 16 * j + cq;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    quad_transpose(v, lane);
                    if (n < a.N) {
                        const float4 b = a.bias ? *(const float4*)(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 x4 = *(const float4*)(xr + n);
                        const float l4[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
                        float s4[4];
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const float l = l4[e4], e = __expf(-fabsf(l)), ope = 1.0f + e;
                            sum += xv[e4] * l - fmaxf(l, 0.0f);      // iwae1.py:111: x l - softplus(l) = x l - max(l, 0) - log(1 + e^-|l|)
                            prod *= ope;
                            s4[e4] = xv[e4] - (l >= 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(ope);
                        }
                        if (a.C && m < a.M) *(float4*)(a.C + (size_t)m * a.ldc + n) = make_float4(s4[0], s4[1], s4[2], s4[3]);
                    }
                }
                sum -= __logf(prod);
                sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);      // the row's four column groups c sit on lanes 4c + p
                if ((r16 >> 2) == 0 && m < a.M) a.part[(size_t)(2 * bx + wn) * a.part_stride + m] = sum;
            }
            return;
        }
        float bias4[4];
        bool nok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 64 * wn + 16 * j + r16;
            nok[j] = n < a.N;
            bias4[j] = (nok[j] && a.bias) ? a.bias[n] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 64 * wm + 16 * i + 4 * q + r;
                const int mc = min(m, a.M - 1);
                const float* xr = a.XB + (size_t)(mc / a.bern_k) * a.bern_X;
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (nok[j]) {
                        const float l = acc[i][j][r] + bias4[j];
                        const float xv = xr[n0 + 64 * wn + 16 * j + r16];
                        const float e = __expf(-fabsf(l)), ope = 1.0f + e;
                        sum += xv * l - (fmaxf(l, 0.0f) + __logf(ope));      // iwae1.py:111
                        if (a.C && m < a.M) a.C[(size_t)m * a.ldc + n0 + 64 * wn + 16 * j + r16] = xv - (l >= 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(ope);
                    }
                }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
                if (r16 == 0 && m < a.M) a.part[(size_t)(2 * bx + wn) * a.part_stride + m] = sum;
            }
        return;
    }
    }
    float* C = a.C + (size_t)bz * a.slab_stride;
    if (a.cvec) {      // every row of C (and of ACT, the bias) is 16-byte aligned, N is a multiple of 4: transposed tiles, 16-byte accesses
        const int p = lane & 3, cq = 4 * (r16 >> 2);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + 16 * TM * wm + 16 * i + 4 * q + p;           // the lane's row behind the transposition
            const float rs = (a.orow_scale && m < a.M) ? a.orow_scale[m] : 1.0f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + 16 * TN * wn + 16 * j + cq;              // its four columns n .. n + 3
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                quad_transpose(v, lane);
                if (n >= a.N) continue;
                if (m >= a.M) {
                    if (a.Cones && m == a.M) *(float4*)(a.Cones + (size_t)bz * a.cones_stride + n) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                if (a.bias) { const float4 b = *(const float4*)(a.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if (a.orow_scale) { v[0] *= rs; v[1] *= rs; v[2] *= rs; v[3] *= rs; }
                if (a.epi == GEMM_EPI_TANH) { v[0] = tanh_f32(v[0]); v[1] = tanh_f32(v[1]); v[2] = tanh_f32(v[2]); v[3] = tanh_f32(v[3]); }
                else if (a.epi == GEMM_EPI_EXP) { v[0] = expf(v[0]) + 1e-6f; v[1] = expf(v[1]) + 1e-6f; v[2] = expf(v[2]) + 1e-6f; v[3] = expf(v[3]) + 1e-6f; }
                else if (a.epi == GEMM_EPI_DTANH) {
                    const float4 y = *(const float4*)(a.ACT + (size_t)m * a.ldact + n);
                    v[0] *= 1.0f - y.x * y.x; v[1] *= 1.0f - y.y * y.y; v[2] *= 1.0f - y.z * y.z; v[3] *= 1.0f - y.w * y.w;
                }
                float4* dst = (float4*)(C + (size_t)m * a.ldc + n);
                if (a.accumulate) { const float4 o = *dst; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
                *dst = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#ifdef IWAE_DENSE_STAMPS
        GS_STAMP(6)      // epilogue
        if (a.stamps && lane == 0) {
            const size_t w_ = (((size_t)bz * gridDim.y + by) * gridDim.x + bx) * NW + wave;
            for (int i_ = 0; i_ < 8; ++i_) a.stamps[w_ * 8 + i_] = gs_sum[i_];
        }
#endif
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + 16 * TN * wn + 16 * j + r16;
            if (n >= a.N) continue;
            const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * TM * wm + 16 * i + 4 * q + r;
                if (m >= a.M) {
                    if (a.Cones && m == a.M) a.Cones[(size_t)bz * a.cones_stride + n] = acc[i][j][r];
                    continue;
                }
                float v = acc[i][j][r] + bias;
                if (a.orow_scale) v *= a.orow_scale[m];
                if (a.epi == GEMM_EPI_TANH) v = tanh_f32(v);
                else if (a.epi == GEMM_EPI_EXP) v = expf(v) + 1e-6f;
                else if (a.epi == GEMM_EPI_DTANH) { const float y = a.ACT[(size_t)m * a.ldact + n]; v *= 1.0f - y * y; }
                float* dst = C + (size_t)m * a.ldc + n;
                *dst = a.accumulate ? *dst + v : v;
            }
        }
}

// ---------------------------------------------------------------------------------
// dec_fwd_f32_kernel (round 4): the WHOLE decoder forward in float32 -- z -> tanh -> tanh -> logits -> log p(x|z) (src/iwae1.py:79-85,111) --
// in one launch, row-block stationary like the bf16 path's decoder kernel.  Before: three launches of the generic GEMM (88 TFLOP/s on the
// 784-wide product, 53 on the 200-wide ones: tiles of 7-13 k-steps pay their prologue, epilogue and a float32 round trip of every activation
// through HBM) = 0.35 of the f32 MFMA peak for the k = 5000 evaluator.
//   * a wave owns 16 data rows through all three layers; a workgroup = 4 waves = 64 rows, 80.9 KB of LDS -> two independent workgroups per
//     CU (two waves per SIMD in different phases: one's epilogue arithmetic runs beside the other's MFMAs);
//   * v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact float32 products): the wave's accumulators are ALL out-feature tiles of its
//     rows (<= 13 tiles = 208 features per pass; the 784 pixels take 4 passes of 13/12 tiles), A operand = its rows' activations from its
//     PRIVATE strip of LDS, B operand = the layer's weights, streamed 16 in-features at a time through two shared LDS slabs by LDS-DMA
//     straight from the float32 master parameters (Keras [in][out] rows: a slab is 16 contiguous row segments);
//   * because a wave holds the whole layer output of its rows in registers, the layer's activations are dead when its k loop ends: the
//     next layer's input OVERWRITES them in place (one 16 x 212 float strip per wave for z, g1, g2 in turn);
//   * lane quad q contracts k = 4j + q in MFMA step j (the instruction's natural order): a slab row pitch of 208 floats puts quads q, q + 1
//     16 banks apart (conflict-free ds_read_b32), and the activations are stored k-permuted inside each 16-block (position 4(k&3) + (k>>2))
//     so that a lane's four k values of a block are ONE ds_read_b128;
//   * epilogues: bias + tanh (one hardware exp2) -> the strip [+ g1 / g2 rows for the backward pass]; output layer: log p(x|z) summed per
//     row in registers over the passes (a wave owns whole rows: no partial sums, no cross-wave reduction) [+ s = x - sigmoid(l)].
// ---------------------------------------------------------------------------------
#ifdef IWAE_DENSE_STAMPS      // diagnostic build (STAMPS=1): per-wave cycle sums of the kernel's phases -> a.stamps[wave][8]
#define DF_STAMP(slot)                                                                 \
    {                                                                                  \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        df_sum[slot] += t_ - df_prev;                                                  \
        df_prev = t_;                                                                  \
    }
#else
#define DF_STAMP(slot)
#endif
#define DF_TP 13                    // out-feature tiles (16 each) a wave accumulates per pass
#define DF_PA 212                   // activation strip pitch in floats (= 4 * 53: rows fall into different bank quads for ds_read_b128)
#define DF_SLAB (16 * 16 * DF_TP)   // floats per weight slab: 16 in-features x 208 out-features
__device__ __forceinline__ void df_glds16(const char* g, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_wave_base) : "memory");
}
__global__ __launch_bounds__(256, 2) void dec_fwd_f32_kernel(DecFwdF32Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem_df[];
#ifdef IWAE_DENSE_STAMPS
    unsigned long long df_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, df_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(df_prev)::"memory");
#endif
    float* strip_all = (float*)smem_df;                                  // [4 waves][16 rows][DF_PA]
    float* slab_all = (float*)(smem_df + 4 * 16 * DF_PA * 4);            // [2][16][208]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 64 + wave * 16;                          // the wave's first data row
    float* strip = strip_all + wave * 16 * DF_PA;
    const int permn = 4 * (n16 & 3) + (n16 >> 2);                        // position of column n16 inside a 16-block of the strip
    const uint32_t slab_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)slab_all;
    const char* zsrc = a.zero + lane * 16;

    // ---- the wave's 16 rows of z into its strip (k-permuted, pad columns and rows >= M zero): float4 requests, all of a lane's in flight
    // before the first LDS store (element by element with a division per element this was 9 % of the kernel: 40 k cycles per wave)
    {
        const int Kp = (a.Din + 15) & ~15, Q = Kp >> 2;                  // quads per row (Kp / 4 <= 52)
        const bool vec = (a.Din & 3) == 0 && (a.ldz & 3) == 0 && (((uintptr_t)a.Z) & 15) == 0;
        float4 zq[13];                                                     // 16 rows x <= 52 quads = <= 832 quads = 13 per lane
        int zpos[13];
#pragma unroll
        for (int it = 0; it < 13; ++it) {
            const int idx = lane + 64 * it;
            zq[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            zpos[it] = -1;
            if (idx < 16 * Q) {
                const int r = idx / Q, c = 4 * (idx - r * Q);
                zpos[it] = r * DF_PA + (c & ~15) + ((c & 15) >> 2);       // element e of the quad (column c + e) goes to + 4 e
                if (m0 + r < a.M && c < a.Din) {
                    const float* src = a.Z + (size_t)(m0 + r) * a.ldz + c;
                    if (vec) zq[it] = *(const float4*)src;
                    else { zq[it].x = src[0]; if (c + 1 < a.Din) zq[it].y = src[1]; if (c + 2 < a.Din) zq[it].z = src[2]; if (c + 3 < a.Din) zq[it].w = src[3]; }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 13; ++it) {
            if (zpos[it] >= 0) {
                strip[zpos[it]] = zq[it].x; strip[zpos[it] + 4] = zq[it].y; strip[zpos[it] + 8] = zq[it].z; strip[zpos[it] + 12] = zq[it].w;
            }
        }
    }
    // the images of the lane's four rows (4q + r): base of their x rows, once (per load this was an integer division by k: 23 % of the kernel)
    const float* xrow[4];
    const int img0 = min(m0, a.M - 1) / a.k, nimg = min(m0 + 15, a.M - 1) / a.k - img0 + 1;      // (wave-uniform)
    const bool x_in_lds = nimg <= 3;
    int xslot[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int im = min(m0 + 4 * q + r, a.M - 1) / a.k;
        xrow[r] = a.XB + (size_t)im * a.X;
        xslot[r] = x_in_lds ? im - img0 : 0;
    }
    float rowsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    DF_STAMP(0)      // z staging

    // one layer: K in-features -> N out-features.  OUT = false: tanh into the strip (+ Gout); OUT = true: Bernoulli log-likelihood
    auto layer = [&](auto out_tag, const float* W, const float* bias, const int K, const int N, float* Gout) {
        constexpr bool OUT = decltype(out_tag)::value;
        const int NT = (N + 15) >> 4, npass = (NT + DF_TP - 1) / DF_TP, TP = (NT + npass - 1) / npass, nkb = (K + 15) >> 4;
        for (int pass = 0; pass < npass; ++pass) {
            const int t0 = pass * TP, cnt = min(TP, NT - t0), c0 = 16 * t0;
            // DMA of a slab: granule g = tid + 256 u (16 bytes) of the [16][208] slab <- W[16 kb + g / 52][c0 + 4 (g % 52)].  A lane's source
            // advances by 16 weight rows per slab: a running pointer and a stride (both zero-line / 0 for granules outside the layer), two
            // vector adds per piece and slab -- as per-slab index arithmetic + selects it was ~15 (3.8 vector instructions per MFMA in the
            // first version's counters: the waves' issue slots, not the matrix pipe, set its pace)
            const char* gcur[4]; unsigned gstep[4]; int grow[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = tid + 256 * u, r = g / (4 * DF_TP), cq = g - r * (4 * DF_TP);
                const bool ok = r < 16 && cq < 4 * cnt && c0 + 4 * cq < N;      // (N is a multiple of 4: a granule is all in or all out)
                grow[u] = ok ? r : (1 << 20);
                gcur[u] = ok ? (const char*)(W + (size_t)r * N + c0 + 4 * cq) : zsrc;
                gstep[u] = ok ? (unsigned)(16 * N * 4) : 0u;
            }
            const bool ragged = (K & 15) != 0;       // the last slab has rows beyond K: they read the zero line
            auto issue = [&](int kb, int buf) {
                const bool last_ragged = ragged && kb == nkb - 1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (256 * u + 64 * wave < 16 * 4 * DF_TP) {          // (wave-uniform: pieces past the slab's 832 granules are never issued)
                        const char* src = (last_ragged && 16 * kb + grow[u] >= K) ? zsrc : gcur[u];
                        df_glds16(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)(slab_lds + (uint32_t)buf * (DF_SLAB * 4) + (uint32_t)(256 * u + 64 * wave) * 16u)));
                        gcur[u] += gstep[u];
                    }
                }
            };
            f32x4v acc[DF_TP];
#pragma unroll
            for (int t = 0; t < DF_TP; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            DF_STAMP(1)      // pass set-up
            __syncthreads();              // every wave is done with both slabs of the pass / layer before (and the strip is written)
            DF_STAMP(2)      // barrier between passes
            issue(0, 0);
            // The epilogue's operands -- the bias of the pass's 208 columns and, for the output layer, x of the wave's images at those columns --
            // come as ONE 16-byte request per lane and segment, made at the top of the pass's last slab and parked in the slab buffer that
            // is free by then (the wave's quarter of it), from where the epilogue reads them with ds_read_b32.  As 13 + 52 dword loads per
            // pass they were 23 % of the kernel: what a load costs a CU is its instruction, not its bytes (phase stamps: ~200 cycles each).
            const int segcol = c0 + 4 * lane;
            const bool segok = lane < 4 * DF_TP && lane < 4 * cnt && segcol < N;
            float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), px[3] = {pb, pb, pb};
            for (int kb = 0; kb < nkb; ++kb) {
                const int buf = kb & 1;
                DF_STAMP(3)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DF_STAMP(4)      // own DMA pieces landed
                __syncthreads();          // slab kb has landed for everyone; everyone has left slab kb - 1
                DF_STAMP(2)      // barrier
                if (kb + 1 < nkb) issue(kb + 1, buf ^ 1);
                if (kb == nkb - 1 && segok) {
                    pb = *(const float4*)(bias + segcol);
                    if constexpr (OUT) {
                        if (x_in_lds) {
#pragma unroll
                            for (int si = 0; si < 3; ++si)
                                if (si < nimg) px[si] = *(const float4*)(a.XB + (size_t)(img0 + si) * a.X + segcol);
                        }
                    }
                }
                DF_STAMP(5)      // DMA issue (+ the epilogue operands' requests in the last slab)
                const float4 av = *(const float4*)(strip + n16 * DF_PA + 16 * kb + 4 * q);      // row n16 of the wave: k = 16 kb + 4 j + q at .j
                const float a4[4] = {av.x, av.y, av.z, av.w};
                const float* sl = slab_all + buf * DF_SLAB + q * (16 * DF_TP) + n16;
                // 52 MFMAs (step j = i / 13, tile t = i % 13), their B operands through an explicit 8-deep read pipeline: left to itself the
                // compiler emitted ds_read2 -> lgkmcnt(0) -> 2 MFMAs per pair, one exposed LDS round trip per 64 cycles of matrix pipe
                constexpr int NM = 4 * DF_TP, PD = 8;
                float bq[PD];
#pragma unroll
                for (int i = 0; i < PD; ++i) bq[i] = sl[(4 * (i / DF_TP)) * (16 * DF_TP) + 16 * (i % DF_TP)];
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    acc[i % DF_TP] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i / DF_TP], bq[i % PD], acc[i % DF_TP], 0, 0, 0);      // (all 13 tiles, branch-free: the slab columns of tiles >= cnt are zeros)
                    if (i + PD < NM) bq[i % PD] = sl[(4 * ((i + PD) / DF_TP)) * (16 * DF_TP) + 16 * ((i + PD) % DF_TP)];
                }
                // the order the scheduler must keep: PD reads up front, then one read behind every MFMA (0x100 = DS read, 0x008 = MFMA)
                __builtin_amdgcn_sched_group_barrier(0x100, PD + 1, 0);      // (+ the A operand's ds_read_b128)
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + PD < NM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                DF_STAMP(6)      // fragment reads + MFMAs
            }
            // park the epilogue operands: the buffer slab nkb - 1 did NOT use is free (everyone left it at the top of the last slab, the next
            // DMA into it comes behind the next pass's barriers); wave-private quarter, so the wave's own LDS order is all the ordering needed
            float* scr = slab_all + (nkb & 1) * DF_SLAB + wave * (16 * 4 * DF_TP);
            if (lane < 4 * DF_TP) {
                *(float4*)(scr + 4 * lane) = pb;
                if constexpr (OUT) {
#pragma unroll
                    for (int si = 0; si < 3; ++si) *(float4*)(scr + 16 * DF_TP * (1 + si) + 4 * lane) = px[si];
                }
            }
            // accumulator tile t: lane (n16, q) reg r = row 4q + r of the wave, out-feature c0 + 16 t + n16
            if constexpr (!OUT) {
#pragma unroll
                for (int t = 0; t < DF_TP; ++t) {
                    if (t < cnt) {
                        const int col = c0 + 16 * t + n16;
                        const float bvt = scr[16 * t + n16];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = col < N ? tanh_f32(acc[t][r] + bvt) : 0.0f;
                            strip[(4 * q + r) * DF_PA + 16 * (t0 + t) + permn] = v;       // (in place: this layer's input is dead, see above)
                            if (Gout && col < N && m0 + 4 * q + r < a.M) Gout[(size_t)(m0 + 4 * q + r) * a.ldg + col] = v;
                        }
                    }
                }
            } else {
                // log p(x|z) = sum_n x l - softplus(l) = sum_n (x - 1/2) l - |l| / 2 - log(1 + e^-|l|)   (iwae1.py:111); the logarithms of a row's
                // 13 columns of this pass as ONE log2 of the product of (1 + e) <= 2^13 -- per logit a multiply, one exp2, four plain instructions
                float s_xl[4] = {0.f, 0.f, 0.f, 0.f}, s_al[4] = {0.f, 0.f, 0.f, 0.f}, prod[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int t = 0; t < DF_TP; ++t) {
                    const int col = c0 + 16 * t + n16;
                    const bool ok = t < cnt && col < N;
                    const float bvt = scr[16 * t + n16];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x1 = x_in_lds ? scr[16 * DF_TP * (1 + xslot[r]) + 16 * t + n16] : (ok ? xrow[r][col] : 0.0f);      // (more than 3 images per wave: k < 8 -- straight from memory)
                        const float l = ok ? acc[t][r] + bvt : 0.0f, xm = ok ? x1 - 0.5f : 0.0f;
                        const float e = __builtin_amdgcn_exp2f(-fabsf(l) * 1.4426950408889634f);      // exp(-|l|)
                        s_xl[r] = fmaf(xm, l, s_xl[r]);
                        s_al[r] += fabsf(l);
                        prod[r] = ok ? fmaf(prod[r], e, prod[r]) : prod[r];
                        if (a.S && ok && m0 + 4 * q + r < a.M)      // training step: s = x - sigmoid(l) = (x - 1/2) - sign(l) (1 / (1 + e) - 1/2)
                            a.S[(size_t)(m0 + 4 * q + r) * a.ldS + col] = xm - __builtin_copysignf(__builtin_amdgcn_rcpf(1.0f + e) - 0.5f, l);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) rowsum[r] += s_xl[r] - 0.5f * s_al[r] - 0.6931471805599453f * __builtin_amdgcn_logf(prod[r]);
            }
        }
    };
    layer(std::false_type{}, a.W1, a.b1, a.Din, a.H, a.G1);
    DF_STAMP(7)
    layer(std::false_type{}, a.W2, a.b2, a.H, a.H, a.G2);
    DF_STAMP(7)
    layer(std::true_type{}, a.W3, a.b3, a.H, a.X, nullptr);
    DF_STAMP(7)      // (epilogues: what is left of a pass behind its last slab)
#ifdef IWAE_DENSE_STAMPS
    if (a.stamps && lane == 0) {
        for (int i_ = 0; i_ < 8; ++i_) a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i_] = df_sum[i_];
    }
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = rowsum[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);      // the 16 columns of a tile sit on lanes n16
        if (n16 == 0 && m0 + 4 * q + r < a.M) a.lpxz[m0 + 4 * q + r] = v;
    }
}
bool dec_fwd_f32_ok(const DecFwdF32Args& a) {
    return a.Din >= 1 && a.Din <= 16 * DF_TP && a.H >= 4 && a.H <= 16 * DF_TP && a.H % 4 == 0 && a.X % 4 == 0 && a.k >= 1 &&
           ((uintptr_t)a.W1 & 15) == 0 && ((uintptr_t)a.W2 & 15) == 0 && ((uintptr_t)a.W3 & 15) == 0 &&
           ((uintptr_t)a.b1 & 15) == 0 && ((uintptr_t)a.b2 & 15) == 0 && ((uintptr_t)a.b3 & 15) == 0 && ((uintptr_t)a.XB & 15) == 0;      // (16-byte segment requests)
}
void launch_dec_fwd_f32(const DecFwdF32Args& a, hipStream_t st) {
    const size_t lds = (size_t)4 * 16 * DF_PA * 4 + (size_t)2 * DF_SLAB * 4;
    hipLaunchKernelGGL(dec_fwd_f32_kernel, dim3((a.M + 63) / 64), dim3(256), lds, st, a);
}

// out[i] = sum over z of slabs[z*stride + i] (fixed order), i < n
// Block = 64 consecutive elements x 4 slab groups (slab z goes to group z & 3), 8 slabs' loads in flight per thread, the groups' sums added in
// group order through LDS: as one thread per element walking all slabs in turn this was a chain of up to 247 dependent round trips on a few
// hundred workgroups (23 us per launch, 14 launches per float32 step: the second largest item of its kernel trace).
__global__ __launch_bounds__(256) void reduce_slabs_f32_kernel(const float* slabs, size_t stride, int nsplit, size_t n, float* out) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + c;
    float s = 0.0f;
    if (i < n) {
        const float* p = slabs + i;
        int z = g;
        for (; z + 28 < nsplit; z += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + 4 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < nsplit; z += 4) s += p[(size_t)z * stride];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < n) out[i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// Round 5: every slab sum of a float32 step in ONE launch.  A step has 14 row-split weight / bias gradients; as 14 launches of
// reduce_slabs_f32_kernel (5.7 us each, latency) they were 76 us of the 1.52 ms step.  The gradients now keep their slabs (each in its own
// region of the slab buffer) and the host queues a job per tensor; this kernel walks the concatenated block ranges.  Same block shape, same
// order of additions per element as reduce_slabs_f32_kernel: bitwise the same sums.
__global__ __launch_bounds__(256) void reduce_slabs_multi_f32_kernel(ReduceSlabsJobs jobs) {
    __shared__ float red[4][64];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].block_begin) ++j;
    const ReduceSlabsJob J = jobs.job[j];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)((int)blockIdx.x - J.block_begin) * 64 + c;
    float s = 0.0f;
    if (i < J.n) {
        const float* p = J.slabs + i;
        int z = g;
        for (; z + 28 < J.nsplit; z += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + 4 * u) * J.stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < J.nsplit; z += 4) s += p[(size_t)z * J.stride];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < J.n) J.out[i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// log p(x|z) per data row from float32 logits: sum_j x_j l_j - softplus(l_j) (iwae1.py:111); one wave per row (image-major rows: b = row / k)
__global__ __launch_bounds__(256) void bern_f32_kernel(const float* logits, size_t ld, const float* x, int X, int M, int k, float* lpxz) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* l = logits + (size_t)row * ld;
    const float* xr = x + (size_t)(row / k) * X;
    float s = 0.0f;
    for (int j = lane; j < X; j += 64) {
        const float v = l[j];
        s += xr[j] * v - (fmaxf(v, 0.0f) + log1pf(expf(-fabsf(v))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) lpxz[row] = s;
}

// in place: logits -> dl = gx[row] * (x - sigmoid(l))   (d loss / d logits)
__global__ __launch_bounds__(256) void dl_f32_kernel(float* logits, size_t ld, const float* x, int X, int M, int k, const float* gx) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * X) return;
    const int row = (int)(idx / X), j = (int)(idx - (size_t)row * X);
    float* p = logits + (size_t)row * ld + j;
    const float v = *p;
    *p = gx[row] * (x[(size_t)(row / k) * X + j] - 1.0f / (1.0f + expf(-v)));
}

// out[row][f] = 1 / (1 + exp(-in)): IWAE.sample's probs (iwae1.py:174)
__global__ __launch_bounds__(256) void sigmoid_f32_kernel(float* v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = 1.0f / (1.0f + expf(-v[i]));
}

// [B*k][X] image-major rows -> the reference's [k][B][X] order (src/iwae1.py:148)
__global__ __launch_bounds__(256) void export_mat_kernel(const float* in, int B, int k, int X, float* out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * k * X) return;
    const int row = (int)(idx / X), j = (int)(idx - (size_t)row * X);
    const int b = row / k, s = row - b * k;
    out[((size_t)s * B + b) * X + j] = in[idx];
}
void launch_export_mat(const float* in, int B, int k, int X, float* out, hipStream_t st) {
    hipLaunchKernelGGL(export_mat_kernel, dim3((unsigned)(((size_t)B * k * X + 255) / 256)), dim3(256), 0, st, in, B, k, X, out);
}

// out[r] = concat(a[r] (na floats), b[r] (nb floats)): the conditional models' inputs (tasks/task05.py:113, :185: concat(x, y), concat(z, y))
__global__ __launch_bounds__(256) void concat_f32_kernel(const float* a, int na, const float* b, int nb, int rows, float* out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int w = na + nb;
    if (idx >= (size_t)rows * w) return;
    const int r = (int)(idx / w), j = (int)(idx - (size_t)r * w);
    out[idx] = j < na ? a[(size_t)r * na + j] : b[(size_t)r * nb + (j - na)];
}
void launch_concat_f32(const float* a, int na, const float* b, int nb, int rows, float* out, hipStream_t st) {
    hipLaunchKernelGGL(concat_f32_kernel, dim3((unsigned)(((size_t)rows * (na + nb) + 255) / 256)), dim3(256), 0, st, a, na, b, nb, rows, out);
}

int g_gemm_f32_dbg = 0;
bool g_gemm_f32_v2 = true;
int g_gemm_f32_ksplit_min_tiles = 1;    // (option f32_ksplit_min_tiles: 8 = round-5 first version: B >= 100 images only)
bool g_gemm_f32_ksplit = true;        // (option f32_no_ksplit = 1: few-row products as one 64-tile launch walking K alone)
int g_gemm_f32_v2_small_min = 1;        // (option f32_gemm_small_min: workgroups from which the 64 x 64 launch takes the v2 loop)
bool g_gemm_f32_v2_small = true;      // (option f32_gemm_small_v1 = 1)
bool g_gemm_f32_w8 = true;       // (iwae_set_option f32_gemm_w4 = 1: no 8-wave tiles)      // (iwae_set_option f32_gemm_v1 = 1: the round-3 loop, for A/B measurements; process-wide)
// Tile choice of the big kernels.  4-wave tiles (1 024 workgroup slots on the chip): 128 x 128, 64 x 224, 224 x 64 -- the candidate with the least padded
// area (ties: 128 x 128).  8-wave tiles (512 slots; 39 % fewer operand bytes per FLOP): 128 x 224 where the 64 x 224 tile won and the rows fill the
// machine, 224 x 128 where 224 x 64 won and the wider tile pads <= 10 % more (N = 784: 7 x 128 = 896 against 13 x 64 = 832).
struct GemmF32Tile { int bm, bn, waves; long tm, tn; };
static GemmF32Tile gemm_f32_pick(int M, int N, bool allow8 = true) {
    const int cand[3][2] = {{128, 128}, {64, 224}, {224, 64}};
    long best = -1;
    GemmF32Tile t = {128, 128, 4, 0, 0};
    for (int c = 0; c < 3; ++c) {
        const long tm = (M + cand[c][0] - 1) / cand[c][0], tn = (N + cand[c][1] - 1) / cand[c][1];
        const long area = tm * cand[c][0] * tn * cand[c][1];
        if (best < 0 || area < best) { best = area; t = {cand[c][0], cand[c][1], 4, tm, tn}; }
    }
    if (allow8 && g_gemm_f32_v2 && g_gemm_f32_w8) {
        if (t.bm == 64 && t.bn == 224 && M >= 128 * 256) { t.bm = 128; t.waves = 8; t.tm = (M + 127) / 128; }
        else if (t.bm == 224 && t.bn == 64) {
            const long tn8 = (N + 127) / 128;
            if (tn8 * 128 * 10 <= t.tn * 64 * 11) { t.bn = 128; t.waves = 8; t.tn = tn8; }
        }
    }
    return t;
}
long gemm_f32_tiles(int M, int N, int tile_mode) {      // output tiles of the kernel launch_gemm_f32 would take (f32_dw sizes its row splits from it)
    if (M > 64 && N > 64) { const GemmF32Tile t = gemm_f32_pick(M, N, tile_mode == 0); return t.tm * t.tn; }
    return (long)((M + 63) / 64) * ((N + 63) / 64);
}
int gemm_f32_slots(int M, int N, int tile_mode) {       // workgroups of that kernel the chip holds at once
    if (tile_mode == 2) return 768;
    if (M > 64 && N > 64) return gemm_f32_pick(M, N, tile_mode == 0).waves == 8 ? 512 : 1024;
    return 1024;
}
bool gemm_f32_takes_big(int M, int N, int nsplit) {
    if (!(M > 64 && N > 64)) return false;
    const GemmF32Tile t = gemm_f32_pick(M, N, false);
    return t.tm * t.tn * nsplit >= 512;      // (448, which lets the 100 x 200 weight gradient in -- 2 tiles x 247 row splits -- measured slower: 79 vs 59 us)
}
void launch_gemm_f32(const GemmF32Args& a0, int nsplit, hipStream_t st) {
    GemmF32Args a = a0;
    a.dbg = g_gemm_f32_dbg;
    // float4 fetches where every quad is 16-byte aligned: base pointer, the non-unit stride and the k-chunk offsets
    const long a_str = a.sak == 1 ? a.sam : a.sak, b_str = a.sbn == 1 ? a.sbk : a.sbn;
    a.avec = (((uintptr_t)a.A & 15) == 0 && a_str % 4 == 0 && (a.sak == 1 || a.sam == 1) && (nsplit == 1 || a.kchunk % 4 == 0)) ? 1 : 0;
    a.bvec = (((uintptr_t)a.B & 15) == 0 && b_str % 4 == 0 && (a.sbn == 1 || a.sbk == 1) && (nsplit == 1 || a.kchunk % 4 == 0)) ? 1 : 0;
    a.cvec = (a.N % 4 == 0 && ((uintptr_t)a.C & 15) == 0 && a.ldc % 4 == 0 && a.slab_stride % 4 == 0 && (!a.bias || ((uintptr_t)a.bias & 15) == 0) &&
              (a.epi != GEMM_EPI_DTANH || (((uintptr_t)a.ACT & 15) == 0 && a.ldact % 4 == 0)) &&
              (!a.Cones || (((uintptr_t)a.Cones & 15) == 0 && a.cones_stride % 4 == 0))) ? 1 : 0;
    // 128 x 128 tiles wherever both extents exceed one 64-tile (the per-sample layers, the weight gradients); the small kernel for the
    // rest (few images, narrow heads: a 128-tile would be mostly padding)
    // -- and only where that still fills the machine: a handful of 128-tiles walking K alone is latency-bound (3 us per k-step)
    const int Mg = a.M + (a.Cones ? 1 : 0);      // (the row of ones)
    if (gemm_f32_takes_big(a.M, a.N, nsplit)) {
        // (the v2 loop wants unit strides on the fast index of each operand -- every caller's are -- and falls back to the old kernel otherwise)
        const bool v2 = g_gemm_f32_v2 && (a.sak == 1 || a.sam == 1) && (a.sbn == 1 || a.sbk == 1);
        GemmF32Tile t = gemm_f32_pick(a.M, a.N, v2 && a.tile_mode == 0);
        if (a.epi == GEMM_EPI_BERN) t = {128, 128, 4, 0, 0};      // (the Bernoulli epilogue's partial sums are per 64-column half of a 128-tile)
        const bool ak = a.sak == 1, bnf = a.sbn == 1;
        if (v2 && a.tile_mode == 2 && t.bm == 224 && !ak && bnf) {      // a weight gradient that shares the CUs with few-row kernels: 3 waves per SIMD, one slot left to them
            const dim3 grid((a.N + t.bn - 1) / t.bn, (Mg + t.bm - 1) / t.bm, nsplit);
            hipLaunchKernelGGL((gemm_f32_v2_kernel<7, 2, 2, 2, false, true, 3>), grid, dim3(256), 0, st, a);
            return;
        }
        if (t.waves == 8 && !((t.bm == 128 && ak) || (t.bm == 224 && !ak && bnf))) t = gemm_f32_pick(a.M, a.N, false);      // (8-wave kernels exist for the orientations the step has)
        const dim3 grid((a.N + t.bn - 1) / t.bn, (Mg + t.bm - 1) / t.bm, nsplit);
        if (t.waves == 8) {
            if (t.bm == 128) {
                if (bnf) hipLaunchKernelGGL((gemm_f32_v2_kernel<2, 7, 4, 2, true, true>), grid, dim3(512), 0, st, a);
                else hipLaunchKernelGGL((gemm_f32_v2_kernel<2, 7, 4, 2, true, false>), grid, dim3(512), 0, st, a);
            } else hipLaunchKernelGGL((gemm_f32_v2_kernel<7, 2, 2, 4, false, true>), grid, dim3(512), 0, st, a);
            return;
        }
#define IWAE_F32_BIG(TM, TN)                                                                                             \
        do {                                                                                                             \
            if (v2) {                                                                                                    \
                if (ak && bnf) hipLaunchKernelGGL((gemm_f32_v2_kernel<TM, TN, 2, 2, true, true>), grid, dim3(256), 0, st, a);  \
                else if (ak) hipLaunchKernelGGL((gemm_f32_v2_kernel<TM, TN, 2, 2, true, false>), grid, dim3(256), 0, st, a);   \
                else if (bnf) hipLaunchKernelGGL((gemm_f32_v2_kernel<TM, TN, 2, 2, false, true>), grid, dim3(256), 0, st, a);  \
                else hipLaunchKernelGGL((gemm_f32_v2_kernel<TM, TN, 2, 2, false, false>), grid, dim3(256), 0, st, a);          \
            }                                                                                                            \
            else if (ak && bnf) hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, true, true>), grid, dim3(256), 0, st, a);     \
            else if (ak) hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, true, false>), grid, dim3(256), 0, st, a);      \
            else if (bnf) hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, false, true>), grid, dim3(256), 0, st, a);     \
            else hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, false, false>), grid, dim3(256), 0, st, a);             \
        } while (0)
        if (t.bm == 64) IWAE_F32_BIG(2, 7);
        else if (t.bm == 224) IWAE_F32_BIG(7, 2);
        else IWAE_F32_BIG(4, 4);
#undef IWAE_F32_BIG
    }
    else {
        const dim3 grid((a.N + 63) / 64, (Mg + 63) / 64, nsplit);
        const bool ak = a.sak == 1, bnf = a.sbn == 1;
        // many 64 x 64 tiles (the 50-wide latent layer on all rows, its weight gradient's row splits): the v2 loop on that tile
        if (g_gemm_f32_v2 && g_gemm_f32_v2_small && (long)grid.x * grid.y * grid.z >= g_gemm_f32_v2_small_min && (a.sak == 1 || a.sam == 1) && (a.sbn == 1 || a.sbk == 1)) {
            if (ak && bnf) hipLaunchKernelGGL((gemm_f32_v2_kernel<2, 2, 2, 2, true, true>), grid, dim3(256), 0, st, a);
            else if (ak) hipLaunchKernelGGL((gemm_f32_v2_kernel<2, 2, 2, 2, true, false>), grid, dim3(256), 0, st, a);
            else if (bnf) hipLaunchKernelGGL((gemm_f32_v2_kernel<2, 2, 2, 2, false, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((gemm_f32_v2_kernel<2, 2, 2, 2, false, false>), grid, dim3(256), 0, st, a);
            return;
        }
        // few workgroups walking a long K: two k-steps per iteration (half as many exposed round trips)
        if ((long)grid.x * grid.y * grid.z < 512 && a.kchunk >= 64) hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, st, a);
    }
}
// C = epi(sum over z of slab[z] + bias) for a K-split product of FEW rows (the encoder's layers on the batch's images: 64 tiles of 64 x 64 walking K alone
// expose a memory round trip per k-step -- 20 us for K = 200, 55 for K = 784; split four ways they are 256 workgroups of 3-13 k-steps and this pass):
// the epilogue of the GEMM kernels (bias, row weight, tanh / exp / tanh', accumulate) on the summed slabs, in slab order (deterministic).
__global__ __launch_bounds__(256) void reduce_epi_f32_kernel(GemmF32Args a, const float* slabs, int nsplit) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)a.N >> 2;      // quads of 4 columns (N % 4 == 0)
    if (q >= (size_t)a.M * per) return;
    const int m = (int)(q / per), n = 4 * (int)(q - (size_t)m * per);
    const float* p = slabs + (size_t)m * a.N + n;
    float4 acc = *(const float4*)p;
    for (int z = 1; z < nsplit; ++z) { const float4 t = *(const float4*)(p + (size_t)z * a.slab_stride); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] + (a.bias ? a.bias[n + e] : 0.0f);
        if (a.orow_scale) x *= a.orow_scale[m];
        if (a.epi == GEMM_EPI_TANH) x = tanh_f32(x);
        else if (a.epi == GEMM_EPI_EXP) x = expf(x) + 1e-6f;
        else if (a.epi == GEMM_EPI_DTANH) { const float y = a.ACT[(size_t)m * a.ldact + n + e]; x *= 1.0f - y * y; }
        float* dst = a.C + (size_t)m * a.ldc + n + e;
        *dst = a.accumulate ? *dst + x : x;
    }
}
// the K split launch_gemm_f32_fewrows would take (1: none)
int gemm_f32_fewrows_split(int M, int N, int K) {
    if (!g_gemm_f32_v2 || !g_gemm_f32_ksplit || M > 4096 || (N & 3) || K < 96) return 1;
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles >= 256 || tiles < g_gemm_f32_ksplit_min_tiles) return 1;
    int ns = (int)std::min<long>(std::min<long>(tiles < 8 ? 16 : 8, (256 + tiles - 1) / tiles), K / 48);
    if (ns < 2) return 1;
    const int kchunk = ((K + ns - 1) / ns + 15) / 16 * 16;
    return (K + kchunk - 1) / kchunk;
}
// slabs: >= split * M * N floats of scratch (16-byte aligned)
void launch_gemm_f32_fewrows(const GemmF32Args& a0, float* slabs, hipStream_t st) {
    const int ns0 = gemm_f32_fewrows_split(a0.M, a0.N, a0.K);
    GemmF32Args g = a0;
    g.kchunk = ((a0.K + ns0 - 1) / ns0 + 15) / 16 * 16;
    const int ns = (a0.K + g.kchunk - 1) / g.kchunk;
    g.C = slabs; g.ldc = a0.N; g.slab_stride = (size_t)a0.M * a0.N; g.bias = nullptr; g.epi = GEMM_EPI_NONE; g.ACT = nullptr; g.accumulate = 0; g.orow_scale = nullptr;
    launch_gemm_f32(g, ns, st);
    GemmF32Args e = a0;
    e.slab_stride = g.slab_stride;
    hipLaunchKernelGGL(reduce_epi_f32_kernel, dim3((unsigned)(((size_t)a0.M * (a0.N >> 2) + 255) / 256)), dim3(256), 0, st, e, slabs, ns);
}
void launch_reduce_slabs_f32(const float* slabs, size_t stride, int nsplit, size_t n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_slabs_f32_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slabs, stride, nsplit, n, out);
}
void launch_reduce_slabs_multi_f32(ReduceSlabsJobs& jobs, hipStream_t st) {
    int blocks = 0;
    for (int j = 0; j < jobs.n; ++j) { jobs.job[j].block_begin = blocks; blocks += (int)((jobs.job[j].n + 63) / 64); }
    if (blocks > 0) hipLaunchKernelGGL(reduce_slabs_multi_f32_kernel, dim3(blocks), dim3(256), 0, st, jobs);
}
void launch_bern_f32(const float* logits, size_t ld, const float* x, int X, int M, int k, float* lpxz, hipStream_t st) {
    hipLaunchKernelGGL(bern_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, st, logits, ld, x, X, M, k, lpxz);
}
void launch_dl_f32(float* logits, size_t ld, const float* x, int X, int M, int k, const float* gx, hipStream_t st) {
    hipLaunchKernelGGL(dl_f32_kernel, dim3((unsigned)(((size_t)M * X + 255) / 256)), dim3(256), 0, st, logits, ld, x, X, M, k, gx);
}
void launch_sigmoid_f32(float* v, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(sigmoid_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, n);
}

}  // namespace iwae
