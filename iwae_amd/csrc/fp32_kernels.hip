// float32 mode (iwae_config.precision = IWAE_PREC_FP32): the reference's arithmetic -- Keras Dense layers in float32
// (src/iwae1.py:31-34,72-75) -- with every GEMM on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit a k-ordered
// fmaf chain, no reduced-precision operands).  Row-major float32 activations in their natural widths, weights read straight
// from the fp32 master parameters in Keras [in, out] order.  This mode exists for parity (scalars rel 1e-5,
// gradients rel 1e-4 against the float64 oracle) and for the k = 5000 evaluator; the bf16 kernels in kernels.hip are the fast path.
#include <hip/hip_runtime.h>
#include <stdint.h>
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

// Tile choice of the big kernel: the candidate with the least padded area (ties: the 128 x 128 tile); returns its workgroup count per K split
static long gemm_f32_pick(int M, int N, int& bm, int& bn) {
    const int cand[3][2] = {{128, 128}, {64, 224}, {224, 64}};
    long best = -1, wgs = 0;
    for (int c = 0; c < 3; ++c) {
        const long tm = (M + cand[c][0] - 1) / cand[c][0], tn = (N + cand[c][1] - 1) / cand[c][1];
        const long area = tm * cand[c][0] * tn * cand[c][1];
        if (best < 0 || area < best) { best = area; bm = cand[c][0]; bn = cand[c][1]; wgs = tm * tn; }
    }
    return wgs;
}
long gemm_f32_tiles(int M, int N) {      // output tiles of the kernel launch_gemm_f32 would take (f32_dw sizes its row splits from it)
    if (M > 64 && N > 64) { int bm, bn; return gemm_f32_pick(M, N, bm, bn); }
    return (long)((M + 63) / 64) * ((N + 63) / 64);
}
bool gemm_f32_takes_big(int M, int N, int nsplit) {
    if (!(M > 64 && N > 64)) return false;
    int bm, bn;
    return gemm_f32_pick(M, N, bm, bn) * nsplit >= 512;      // (448, which lets the 100 x 200 weight gradient in -- 2 tiles x 247 row splits -- measured slower: 79 vs 59 us)
}
void launch_gemm_f32(const GemmF32Args& a0, int nsplit, hipStream_t st) {
    GemmF32Args a = a0;
    // float4 fetches where every quad is 16-byte aligned: base pointer, the non-unit stride and the k-chunk offsets
    const long a_str = a.sak == 1 ? a.sam : a.sak, b_str = a.sbn == 1 ? a.sbk : a.sbn;
    a.avec = (((uintptr_t)a.A & 15) == 0 && a_str % 4 == 0 && (a.sak == 1 || a.sam == 1) && (nsplit == 1 || a.kchunk % 4 == 0)) ? 1 : 0;
    a.bvec = (((uintptr_t)a.B & 15) == 0 && b_str % 4 == 0 && (a.sbn == 1 || a.sbk == 1) && (nsplit == 1 || a.kchunk % 4 == 0)) ? 1 : 0;
    // 128 x 128 tiles wherever both extents exceed one 64-tile (the per-sample layers, the weight gradients); the small kernel for the
    // rest (few images, narrow heads: a 128-tile would be mostly padding)
    // -- and only where that still fills the machine: a handful of 128-tiles walking K alone is latency-bound (3 us per k-step)
    const int Mg = a.M + (a.Cones ? 1 : 0);      // (the row of ones)
    if (gemm_f32_takes_big(a.M, a.N, nsplit)) {
        int bm = 128, bn = 128;
        if (a.epi != GEMM_EPI_BERN) gemm_f32_pick(a.M, a.N, bm, bn);      // (the Bernoulli epilogue's partial sums are per 64-column half of a 128-tile)
        const dim3 grid((a.N + bn - 1) / bn, (Mg + bm - 1) / bm, nsplit);
        const bool ak = a.sak == 1, bnf = a.sbn == 1;
#define IWAE_F32_BIG(TM, TN)                                                                                             \
        do {                                                                                                             \
            if (ak && bnf) hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, true, true>), grid, dim3(256), 0, st, a);     \
            else if (ak) hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, true, false>), grid, dim3(256), 0, st, a);      \
            else if (bnf) hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, false, true>), grid, dim3(256), 0, st, a);     \
            else hipLaunchKernelGGL((gemm_f32_big_kernel<TM, TN, false, false>), grid, dim3(256), 0, st, a);             \
        } while (0)
        if (bm == 64) IWAE_F32_BIG(2, 7);
        else if (bm == 224) IWAE_F32_BIG(7, 2);
        else IWAE_F32_BIG(4, 4);
#undef IWAE_F32_BIG
    }
    else {
        const dim3 grid((a.N + 63) / 64, (Mg + 63) / 64, nsplit);
        // few workgroups walking a long K: two k-steps per iteration (half as many exposed round trips)
        if ((long)grid.x * grid.y * grid.z < 512 && a.kchunk >= 64) hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, st, a);
    }
}
void launch_reduce_slabs_f32(const float* slabs, size_t stride, int nsplit, size_t n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_slabs_f32_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slabs, stride, nsplit, n, out);
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
