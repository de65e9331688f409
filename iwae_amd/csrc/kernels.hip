// Hand-written gfx950 (CDNA4, MI355X) kernels for the IWAE train / eval step.
// Layout rules: layout.h.  Kernel -> reference mapping (file:line in /root/reference):
//   dense_kernel<EPI_TANH|EPI_HEAD>  BasicBlock.call          src/iwae1.py:36-44 (iwae2.py:37-45)
//   sample_kernel                    qzx.sample + log-probs   src/iwae1.py:59, :107-109
//   dense_kernel<EPI_BERN>           decoder out + Bernoulli  src/iwae1.py:74-75, :83, :111
//   lse_kernel                       log_w, logmeanexp, eq14  src/iwae1.py:113-139, src/utils.py:6-8
//   out_bwd_kernel / dense<EPI_DX> / wgrad_kernel / latent_bwd_kernel
//                                    tape.gradient(loss, w)   src/iwae1.py:155-159
//   adam_kernel                      Adam(eps=1e-4)           main.py:93, src/iwae1.py:160
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <algorithm>
#include "kernels.h"
#include "layout.h"

#ifdef IWAE_DIAG
#define WG_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define WG_DBG(a, bit) false
#endif
namespace iwae {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LOG2PI_F 1.8378770664093453f

// ---------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;   // hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-safe)
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_at(const uint4& u, int j) {
    const uint32_t w = (j < 2) ? u.x : (j < 4) ? u.y : (j < 6) ? u.z : u.w;
    return (j & 1) ? bfhi(w) : bflo(w);
}
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f
// raw v_exp_f32 / v_log_f32 (base 2, ~1 ulp, no denormal fix-up code: arguments here never need it)
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_raw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float tanh_fast(float x) {
    // 1 - 2/(e^{2x}+1): exact limits at +-inf, abs error ~1e-7 (the result is rounded to bf16 anyway)
    return 1.0f - 2.0f * rcp_fast(exp2_raw(x * (2.0f * LOG2E_F)) + 1.0f);
}
__device__ __forceinline__ float sigmoid_fast(float l) { return rcp_fast(1.0f + exp2_raw(-l * LOG2E_F)); }

// async global -> LDS copy (LDS-DMA), 16 B per lane; LDS destination = M0 (wave-uniform byte address)
// + lane*16.  Issued through inline asm on purpose: with the builtin, hipcc cannot prove that the
// DMA does not alias later ds_reads and drains it (s_waitcnt vmcnt(0)) before every k-step, which
// serialises the weight prefetch with the MFMAs.  Completion is ordered by the explicit
// "s_waitcnt vmcnt(0)" + barrier at the top of each group (cdna_hip_programming.md 5.7).
__device__ __forceinline__ uint32_t lds_addr_of(const char* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void glds16(const char* g, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_wave_base) : "memory");
}

// stream `bytes` (multiple of 1 KiB) of an A-image into LDS with the whole workgroup (wave must be
// wave-uniform, e.g. readfirstlane(threadIdx.x >> 6))
template <int NWAVES>
__device__ __forceinline__ void stage_image(const char* src, char* lds, int bytes, int wave, int lane) {
    const uint32_t base = lds_addr_of(lds);
    for (int off = wave * 1024; off < bytes; off += NWAVES * 1024)
        glds16(src + off + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)off)));
}

// Group boundary wait.  The builtin tells hipcc's waitcnt pass that every vector-memory op it knows
// about has retired (otherwise it re-waits vmcnt(0) in front of the first MFMA of the group, which
// also drains the LDS-DMA just issued for the NEXT group); the asm form is the one that must stay:
// it also covers the asm-issued LDS-DMA, which the pass cannot see.
__device__ __forceinline__ void wait_all_vmem() {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A-fragment stream with P ds_read_b128 in flight: rd(i) -> uint4, use(i, frag).  Without this the
// compiler emits read -> lgkmcnt(0) -> 2 MFMA per fragment and a lone wave pays the LDS latency
// (~64-128 cycles) for every 32 cycles of MFMA.
template <int N, int P, class RD, class USE, class SIDE>
__device__ __forceinline__ void lds_pipeline(RD rd, USE use, SIDE side) {
    uint4 av[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
        if (i < N) av[i] = rd(i);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        use(i, av[i % P]);
        if (i + P < N) av[i % P] = rd(i + P);
        side(i);       // e.g. one LDS-DMA piece of the NEXT group: drains under the MFMAs instead of in front of them
    }
}
template <int N, int P, class RD, class USE>
__device__ __forceinline__ void lds_pipeline(RD rd, USE use) {
    lds_pipeline<N, P>(rd, use, [](int) {});
}

// ---------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller.  counter = (row_lo, row_hi, (stream<<24)|d4, step), key = seed
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // both halves of a 32 x 32 product from ONE v_mad_u64_u32 (as __umulhi + a 32-bit multiply the compiler issued two quarter-rate multiplies each)
        unsigned long long p0, p1;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c0), "v"(0xD2511F53u) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c2), "v"(0xCD9E8D57u) : "vcc");
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void normal4(uint64_t grow, uint32_t d4, uint32_t stream, uint32_t step, uint64_t seed, float n[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (stream << 24) | d4, step, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float s24 = 5.9604644775390625e-08f;   // 2^-24
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * s24, u1 = ((float)(r[1] >> 8) + 0.5f) * s24;
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * s24, u3 = ((float)(r[3] >> 8) + 0.5f) * s24;
    // v_log/v_sqrt/v_sin/v_cos (v_sin/v_cos take revolutions: sin(2*pi*u) = v_sin(u)); abs error ~1e-6
    const float ra = __builtin_amdgcn_sqrtf(-2.0f * __logf(u0)), rb = __builtin_amdgcn_sqrtf(-2.0f * __logf(u2));
    n[0] = ra * __builtin_amdgcn_cosf(u1); n[1] = ra * __builtin_amdgcn_sinf(u1);
    n[2] = rb * __builtin_amdgcn_cosf(u3); n[3] = rb * __builtin_amdgcn_sinf(u3);
}
// 4 consecutive eps values for features 4*d4 .. 4*d4+3 of data row (b,s)
__device__ __forceinline__ void eps4(const EpsSrc& e, int b, int s, int row, int d4, int D, float n[4]) {
    if (e.cache) {
        const float4 v = *(const float4*)(e.cache + (size_t)row * e.ldC + 4 * d4);
        n[0] = v.x; n[1] = v.y; n[2] = v.z; n[3] = v.w;
    } else if (e.user) {
        const float* p = e.user + ((size_t)s * e.B + b) * D + 4 * d4;
#pragma unroll
        for (int i = 0; i < 4; ++i) n[i] = (4 * d4 + i < D) ? p[i] : 0.0f;
    } else {
        const uint64_t grow = e.k_total ? e.row_offset + (uint64_t)b * (uint64_t)e.k_total + (uint64_t)(e.s_off + s) : e.row_offset + (uint64_t)row;
        normal4(grow, (uint32_t)d4, e.stream, e.step, e.seed, n);
    }
}

// One 32-feature step t of z = mu + sigma*eps for the lane's (row, quad): the lane's 8 features of P-layout chunk 4t+q (features
// 32t+4q..+3 and 32t+16+4q..+3) and their contributions to the row's log-densities (sample_kernel; block_fwd_kernel's sampling mode)
__device__ __forceinline__ void sample_step(const SampleArgs& a, int t, int q, int rowc, int b, int s, const float* hd, float z8[8], float& lp, float& lq, float& lq2) {
    // Round 5: the density arithmetic in the decoder kernel's new form -- a density scored at its own sample has (z - mu)/sigma = eps, the
    // chunks that straddle or lie beyond the latent width take a wave-uniform masked path instead of a branch per element, and the
    // log-scales are summed as log2 of pair products (one v_log per 2 elements; __logf was ~12 instructions per element).  The k = 5000
    // evaluator's sampling pass is this function behind Philox + Box-Muller: 107 us per 520 k rows before.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int f0 = 32 * t + 16 * h + 4 * q;
        float e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float4 mu4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sg4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        float4 pm4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ps4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);     // prior: N(0,1) unless a head is given
        if (f0 < a.D) {
            eps4(a.eps, b, s, rowc, f0 >> 2, a.D, e);
            mu4 = *(const float4*)(hd + f0);
            sg4 = *(const float4*)(hd + a.Dp + f0);
            if (a.prior_head) {      // learned conditional prior p(z|y) of tasks/task04.py:124-130, one head per image
                pm4 = *(const float4*)(a.prior_head + (size_t)b * a.ldH + f0);
                ps4 = *(const float4*)(a.prior_head + (size_t)b * a.ldH + a.Dp + f0);
            }
        }
        float muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, sgv[4] = {sg4.x, sg4.y, sg4.z, sg4.w};
        float pmv[4] = {pm4.x, pm4.y, pm4.z, pm4.w}, psv[4] = {ps4.x, ps4.y, ps4.z, ps4.w};
        float cnt = 4.0f;                                     // valid features of this chunk
        const bool ragged = 32 * t + 16 * h + 16 > a.D;      // wave-uniform where t is (sample_kernel, block_fwd_kernel's sampling mode)
        if (ragged) {
            cnt = (float)max(0, min(4, a.D - f0));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = f0 + i < a.D;
                e[i] = in ? e[i] : 0.0f; muv[i] = in ? muv[i] : 0.0f; sgv[i] = in ? sgv[i] : 1.0f;
                pmv[i] = in ? pmv[i] : 0.0f; psv[i] = in ? psv[i] : 1.0f;
            }
        }
        float sz = 0.0f, se = 0.0f, su = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float z = muv[i] + sgv[i] * e[i];                  // iwae1.py:59 (pad features: exactly 0)
            se = fmaf(e[i], e[i], se);                               // iwae1.py:109: (z - mu)/sigma = eps
            if (a.prior_head) {
                const float up = (z - pmv[i]) * __builtin_amdgcn_rcpf(psv[i]);
                sz = fmaf(up, up, sz);                               // task04.py:130
            } else sz = fmaf(z, z, sz);                              // iwae1.py:107
            if (a.lq_dreg) {                                         // tasks/task02.py:63-65: scale sigma + 1e-6
                const float u2 = (sgv[i] * __builtin_amdgcn_rcpf(sgv[i] + 1e-6f)) * e[i];
                su = fmaf(u2, u2, su);
            }
            z8[4 * h + i] = z;
        }
        if (a.cond && ragged) {      // decoder input = concat(z, y): the condition sits in the pad features behind D
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (f0 + i >= a.D && f0 + i < a.D + a.C) z8[4 * h + i] = a.cond[(size_t)b * a.C + (f0 + i - a.D)];
        }
        const float c0 = -0.5f * LOG2PI_F * cnt;
        lp += fmaf(-0.5f, sz, c0) - (a.prior_head ? LN2_F * (log2_raw(psv[0] * psv[1]) + log2_raw(psv[2] * psv[3])) : 0.0f);
        lq += fmaf(-0.5f, se, c0) - LN2_F * (log2_raw(sgv[0] * sgv[1]) + log2_raw(sgv[2] * sgv[3]));
        if (a.lq_dreg) {
            const float s20 = sgv[0] + 1e-6f, s21 = sgv[1] + 1e-6f, s22 = sgv[2] + 1e-6f, s23 = sgv[3] + 1e-6f;
            const float m0 = ragged && !(f0 + 0 < a.D) ? 1.0f : s20, m1 = ragged && !(f0 + 1 < a.D) ? 1.0f : s21;
            const float m2 = ragged && !(f0 + 2 < a.D) ? 1.0f : s22, m3 = ragged && !(f0 + 3 < a.D) ? 1.0f : s23;
            lq2 += fmaf(-0.5f, su, c0) - LN2_F * (log2_raw(m0 * m1) + log2_raw(m2 * m3));
        }
    }
}

// ---------------------------------------------------------------------------------
// Bernoulli log-likelihood of the lane's 8 logits of one 32-pixel step (accumulator tiles a0: pixels f0..f0+3,
// a1: f0+16..f0+19, bias already in), x = the lane's 8 bf16 pixel values:
//   returns sum_j x_j l_j - softplus(l_j)  (iwae1.py:111);  KEEP: sp = bf16 of s_j = x_j - sigmoid(l_j), the gradient of that
//   sum wrt l_j, which the backward kernels read instead of recomputing the logits.
// Per logit: unpack x, x-1/2, mul, exp2, add, mul and two running sums (+ rcp, sub, copysign, sub, half a pack for s).
// With e = exp(-|l|) and max(l,0) = (l + |l|)/2:
//   x*l - softplus(l) = (x - 1/2)*l - |l|/2 - ln2*log2(1+e),   sigmoid(l) - 1/2 = copysign(1/(1+e) - 1/2, l),
// and sum_j log2(1+e_j) = log2(prod_j (1+e_j)): the factors lie in (1,2], so 8 logits cost seven multiplies and ONE
// v_log instead of eight quarter-rate v_log.  MASKED: pixels >= Xdim (pads, l = 0 exactly) contribute nothing.
// Measured (phase stamps, ablations): ~80 issue cycles per logit with s kept, transcendentals at 16 each -- this
// epilogue, not the MFMAs, the LDS stream or the stores, is what the Bernoulli kernel's time is made of.
// ---------------------------------------------------------------------------------
template <bool MASKED, bool KEEP>
__device__ __forceinline__ float bern8(const f32x4& a0, const f32x4& a1, const uint4& xb, int f0, int Xdim, uint4& sp) {
    float s_xl = 0.0f, s_al = 0.0f, prod = 1.0f, sv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float l = (j < 4) ? a0[j & 3] : a1[j & 3];
        const float xm = bf_at(xb, j) - 0.5f;
        const float ope = 1.0f + exp2_raw(-fabsf(l * LOG2E_F));      // 1 + exp(-|l|)
        const bool in = !MASKED || (f0 + 16 * (j >> 2) + (j & 3) < Xdim);
        prod *= in ? ope : 1.0f;
        s_al += fabsf(l);
        s_xl = fmaf(xm, l, s_xl);
        if (KEEP) {
            const float h = __builtin_copysignf(rcp_fast(ope) - 0.5f, l);    // sigmoid(l) - 1/2
            sv[j] = in ? xm - h : 0.0f;
        }
    }
    if (KEEP) sp = make_uint4(pack2(sv[0], sv[1]), pack2(sv[2], sv[3]), pack2(sv[4], sv[5]), pack2(sv[6], sv[7]));
    return s_xl - 0.5f * s_al - LN2_F * log2_raw(prod);
}

// ---------------------------------------------------------------------------------
// dense_kernel: Y^T[out][rows] = W^T-image x X^T, one wave = 32 data rows (2 column groups of
// 16, rows interleaved r0+2*rho+g).
// Weights (and the bias of the group, as a trailing 1 KiB block) stream through LDS one
// 64-out-feature group (x <=8 k-steps) at a time, double buffered with global_load_lds; the
// data operand lives in registers for K <= 256.  Software pipeline per group u:
//   wait+barrier | DMA weights(u+1) | stores of epilogue(u-1) | loads for epilogue(u+1) |
//   MFMA(u) | epilogue math(u)
// so neither load latency nor store acknowledgement sits in front of a barrier.
// ---------------------------------------------------------------------------------
#define DENSE_UNIT 33792   // 8 k-steps x 4 KiB + 1 KiB bias block

// diagnostic build only (./build.sh with STAMPS=1): per-phase s_memtime sums per wave -> a.stamps[wave][8]
#ifdef IWAE_DENSE_STAMPS
#define DS_STAMP(slot)                                                                 \
    {                                                                                  \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        ds_sum[slot] += t_ - ds_prev;                                                  \
        ds_prev = t_;                                                                  \
    }
#else
#define DS_STAMP(slot)
#endif

// G = 16-row column groups per wave.  G = 2: 4 waves x 32 rows, every weight fragment read from LDS feeds two MFMAs
// (the LDS-frugal shape).  G = 1: 8 waves x 16 rows in <= 128 registers, i.e. FOUR waves per SIMD with two workgroups
// per CU: the epilogues here are transcendental-bound on the wave's own issue stream (exp/log/rcp at quarter rate), and
// the only way to fill a SIMD's transcendental unit is more resident waves.
// ZIN: the data operand is not read but MADE: the layer's input is the reparameterised sample z = mu + sigma*eps of its rows
// (first decoder layer of a training step).  The wave draws its fragments from the encoder head of the row's image and the
// step's noise, stores them (they ARE the P-layout rows of z, which the weight gradient needs) and sums the row's prior and
// posterior log-densities on the way -- the separate sampling pass over the rows (42 MB, 14 us) disappears.
template <int EPI, int KTC, int G, int NWV = (G == 2 ? 4 : 8), bool ZIN = false>   // KTC > 0: compile-time k-step count (<= 8, single window), straight-line MFMA phase; NWV waves
__global__ __launch_bounds__(64 * NWV, G == 1 ? 4 : ((EPI == EPI_BERN && KTC > 0) ? 2 : 1)) void dense_kernel(DenseArgs a) {   // 2nd = waves per SIMD (the run-time-K Bernoulli fallback -- hidden widths without an instantiation -- spilled 50 registers at two)
#ifdef IWAE_DENSE_STAMPS
    unsigned long long ds_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ds_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ds_prev)::"memory");
#endif
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = (blockIdx.x * NWV + wave) * (16 * G);
    int row[G], rowc[G];
    bool valid[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { row[g] = r0 + G * rho + g; valid[g] = row[g] < a.M; rowc[g] = min(row[g], a.M - 1); }
    // loads are issued unconditionally from a clamped row and zeroed by a select afterwards: a
    // branch per load would put a vmcnt wait behind every one of them
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    const int KT = KTC ? KTC : a.KT;
    const int nkw = KTC ? 1 : ((KT + 7) >> 3);
    const int mg0 = blockIdx.y * a.mg_per_block;
    const int mg1 = min(a.MG, mg0 + a.mg_per_block);
    const int nunits = (mg1 - mg0) * nkw;
    const size_t gbytes = img_mg_group_bytes(KT);
    constexpr bool kPacked = (EPI == EPI_TANH || EPI == EPI_DX);     // bf16 P/T outputs
    constexpr bool kPre = (EPI == EPI_DX || EPI == EPI_BERN);        // epilogue reads a global operand
    constexpr bool kBiasInit = (EPI == EPI_BERN) && KTC > 0;         // accumulators start at the bias (single k-window: bias block is there)

    uint4 bfr[8][G];
    auto load_b_into = [&](int kw, uint4 (&dst)[8][G]) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (KTC ? ks < KTC : kw * 8 + ks < KT) {
                    v = *(const uint4*)(a.X + (size_t)rowc[g] * a.ldX + (kw * 8 + ks) * 32 + q * 8);
                    if (!valid[g]) v = make_uint4(0, 0, 0, 0);
                }
                dst[ks][g] = v;
            }
        }
    };
    auto load_b = [&](int kw) { load_b_into(kw, bfr); };
    // K > 256 (several 8-k-step windows, e.g. the 784-pixel input layer): the next window's data rows are fetched while
    // this window's MFMAs run, instead of in front of its barrier
    uint4 bfr_n[KTC ? 1 : 8][KTC ? 1 : G];
    auto stage = [&](int unit, int buf) {
        const int mg = mg0 + unit / nkw, kw = unit % nkw;
        const int nks = min(8, KT - kw * 8);
        const int bytes = nks * 4096 + ((kw == nkw - 1) ? 1024 : 0);   // last window carries the bias block
        stage_image<NWV>(a.img + (size_t)mg * gbytes + (size_t)kw * 8 * 4096, smem + buf * DENSE_UNIT, bytes, wave, lane);
    };
    // compile-time shapes: the group is NPC 1 KiB DMA pieces, wave w moves pieces w, w+4, ...; they are issued
    // one at a time between the MFMAs of the previous group (the per-CU L2->LDS rate, ~60-70 GB/s, is what a
    // lump of 29 pieces in front of the MFMAs would wait for)
    constexpr int NPC = (KTC ? KTC : 1) * 4 + 1, NIDXC = (NPC + NWV - 1) / NWV;
    auto dma_piece = [&](int unit, int buf, int idx) {
        const int p = wave + NWV * idx;               // wave-uniform
        if (p < NPC)
            glds16(a.img + (size_t)(mg0 + unit) * gbytes + (size_t)p * 1024 + lane * 16,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * DENSE_UNIT) + (uint32_t)p * 1024u)));
    };

    // EPI_BERN state
    float rowacc[G];
    int bidx[G], sidx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { rowacc[g] = 0.0f; bidx[g] = 0; sidx[g] = 0; }
    if (EPI == EPI_BERN) {
#pragma unroll
        for (int g = 0; g < G; ++g) { bidx[g] = valid[g] ? row[g] / a.k : 0; sidx[g] = valid[g] ? row[g] - bidx[g] * a.k : 0; }
    }
    // lane-constant byte offsets (32-bit) next to wave-uniform 64-bit bases

    auto load_pre = [&](int mg, uint4 (&pre)[2][G]) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                pre[p][g] = make_uint4(0, 0, 0, 0);
                const int fbase = 64 * mg + 32 * p;      // wave-uniform guards only
                if (EPI == EPI_DX && fbase < a.Np32) pre[p][g] = *(const uint4*)(a.ACT + (size_t)rowc[g] * a.ldACT + fbase + 8 * q);
                if (EPI == EPI_BERN && fbase < a.ldXB) pre[p][g] = *(const uint4*)(a.XB + (size_t)bidx[g] * a.ldXB + fbase + 8 * q);
                if (!valid[g]) pre[p][g] = make_uint4(0, 0, 0, 0);
            }
    };

    // deferred bf16 stores of the previous group
    uint4 stP[2][G];
    int st_mg = -1;
    auto emit_stores = [&]() {
        if (!(kPacked || EPI == EPI_BERN) || st_mg < 0) return;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fbase = 64 * st_mg + 32 * p;
            if (fbase < a.Np32) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (valid[g]) *(uint4*)(a.YP + (size_t)row[g] * a.ldYP + fbase + 8 * q) = stP[p][g];
            }
        }
        st_mg = -1;
    };

    // a.stage_all (run-time K path, <= 4 units per block, i.e. small row counts where a block owns ONE out-feature group):
    // all windows' weights are requested up front into their own LDS buffers -- the window loop then waits once for the DMA
    // instead of once per window (it was 45 % wait + 37 % issue at B = 1024)
    const bool stage_all = !KTC && a.stage_all && nunits <= 4;
    if (stage_all) {
        for (int u = 0; u < nunits; ++u) stage(u, u);
    } else if (nunits > 0) stage(0, 0);
    if constexpr (ZIN) {
        static_assert(KTC > 0, "the sampled-input mode is a single-window launch");
        float lp[G], lq[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            lp[g] = 0.0f; lq[g] = 0.0f;
            const int b = rowc[g] / a.k;
            const float* hd = a.zhead + (size_t)b * a.ldZH;
            const float* er = a.zeps + (size_t)rowc[g] * a.zldE;
#pragma unroll
            for (int ks = 0; ks < KTC; ++ks) {
                float z8[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int f0 = 32 * ks + 16 * h + 4 * q;          // the lane's features f0..f0+3 of this k-step (P order)
                    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f), mu4 = e4, sg4 = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (f0 < a.zD) {
                        e4 = *(const float4*)(er + f0);
                        mu4 = *(const float4*)(hd + f0);
                        sg4 = *(const float4*)(hd + a.zDp + f0);
                    }
                    const float ev[4] = {e4.x, e4.y, e4.z, e4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, sgv[4] = {sg4.x, sg4.y, sg4.z, sg4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float z = 0.0f;
                        if (f0 + i < a.zD) {
                            z = fmaf(sgv[i], ev[i], muv[i]);                              // iwae1.py:59
                            lp[g] += -0.5f * z * z - 0.5f * LOG2PI_F;                     // iwae1.py:107
                            lq[g] += -0.5f * ev[i] * ev[i] - 0.5f * LOG2PI_F - __logf(sgv[i]);   // iwae1.py:109: (z - mu)/sigma IS eps
                        }
                        z8[4 * h + i] = z;
                    }
                }
                const uint4 frag = make_uint4(pack2(z8[0], z8[1]), pack2(z8[2], z8[3]), pack2(z8[4], z8[5]), pack2(z8[6], z8[7]));
                bfr[ks][g] = valid[g] ? frag : make_uint4(0, 0, 0, 0);
                if (valid[g]) *(uint4*)(a.ZPout + (size_t)row[g] * a.ldX + ks * 32 + q * 8) = frag;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float v0 = lp[g], v1 = lq[g];
            v0 += __shfl_xor(v0, 16); v0 += __shfl_xor(v0, 32);
            v1 += __shfl_xor(v1, 16); v1 += __shfl_xor(v1, 32);
            if (q == 0 && valid[g]) { a.zlp[row[g]] = v0; a.zlq[row[g]] = v1; }
        }
    } else if (nkw == 1) load_b(0);
    uint4 pre[2][G], pre_n[2][G];
    if (kPre && mg0 < mg1) load_pre(mg0, pre);
    DS_STAMP(0)      // prologue

    for (int mg = mg0; mg < mg1; ++mg) {
        f32x4 acc[4][G];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

        const char* lbias = smem;
        for (int kw = 0; kw < nkw; ++kw) {
            const int unit = (mg - mg0) * nkw + kw, buf = stage_all ? unit : (unit & 1);
            if constexpr (KTC == 0) if (nkw > 1) {
                if (unit == 0) {
                    load_b(kw);
                } else {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                        for (int g = 0; g < G; ++g) bfr[ks][g] = bfr_n[ks][g];
                }
            }
            wait_all_vmem();
            DS_STAMP(1)      // vmcnt wait
            __syncthreads();
            DS_STAMP(2)      // barrier
            const bool more = unit + 1 < nunits;
            if (!KTC && more && !stage_all) stage(unit + 1, buf ^ 1);
            if constexpr (KTC == 0) { if (nkw > 1 && more) load_b_into((kw + 1) % nkw, bfr_n); }
            if (kw == 0) {
                emit_stores();
                if (kPre && mg + 1 < mg1) load_pre(mg + 1, pre_n);
            }
            DS_STAMP(3)      // issue stores / prefetch loads
            const int nks = KTC ? KTC : min(8, KT - kw * 8);
            const char* lb = smem + buf * DENSE_UNIT + a_off;
            lbias = smem + buf * DENSE_UNIT + nks * 4096;
            if (kBiasInit) {      // bias of features 16t + 4q + i is what accumulator register i of tile t starts from
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 b4 = *(const float4*)(lbias + (16 * t + 4 * q) * 4);
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[t][g] = (f32x4){b4.x, b4.y, b4.z, b4.w};
                }
            }
            if (KTC) {
                constexpr int NF = (KTC ? KTC : 1) * 4, STEP = NF / NIDXC > 0 ? NF / NIDXC : 1;
                lds_pipeline<NF, 8>(
                    [&](int i) { return *(const uint4*)(lb + i * 1024); },
                    [&](int i, const uint4& av) {
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[i & 3][g] = mfma16(av, bfr[i >> 2][g], acc[i & 3][g]);
                    },
                    [&](int i) { if (more && i % STEP == 0 && i / STEP < NIDXC) dma_piece(unit + 1, buf ^ 1, i / STEP); });
                if (more) {
#pragma unroll
                    for (int idx = (NF + STEP - 1) / STEP; idx < NIDXC; ++idx) dma_piece(unit + 1, buf ^ 1, idx);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks < nks) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const uint4 av = *(const uint4*)(lb + (ks * 4 + t) * 1024);
#pragma unroll
                            for (int g = 0; g < G; ++g) acc[t][g] = mfma16(av, bfr[ks][g], acc[t][g]);
                        }
                    }
                }
            }
        }

        DS_STAMP(4)      // MFMA phase
        // ---------------- epilogue math for out-features 64*mg .. 64*mg+63 ----------------
        float4 bias4[4];   // bias of features 16t + 4q + i, from the image's bias block
#pragma unroll
        for (int t = 0; t < 4; ++t) bias4[t] = *(const float4*)(lbias + (16 * t + 4 * q) * 4);
        auto bias_of = [&](int t, int i) { return i == 0 ? bias4[t].x : i == 1 ? bias4[t].y : i == 2 ? bias4[t].z : bias4[t].w; };

        if (kPacked) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[G][8];
#pragma unroll
                for (int g = 0; g < G; ++g) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = acc[2 * p + (j >> 2)][g][j & 3];
                        if (EPI == EPI_TANH) {
                            v[g][j] = tanh_fast(x + bias_of(2 * p + (j >> 2), j & 3));
                        } else {
                            const float y = bf_at(pre[p][g], j);
                            v[g][j] = x * (1.0f - y * y);
                        }
                    }
                    stP[p][g] = make_uint4(pack2(v[g][0], v[g][1]), pack2(v[g][2], v[g][3]), pack2(v[g][4], v[g][5]), pack2(v[g][6], v[g][7]));
                }
            }
            st_mg = mg;
        } else if (EPI == EPI_HEAD || EPI == EPI_F32 || EPI == EPI_SIGMOID) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f0 = 64 * mg + 16 * t + 4 * q;
                if (f0 < a.ldYF) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        float op[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x = acc[t][g][i];
                            if (EPI != EPI_F32) x += bias_of(t, i);
                            if (EPI == EPI_HEAD && f0 + i >= a.split) x = exp2_raw(x * LOG2E_F) + 1e-6f;   // iwae1.py:34,42
                            if (EPI == EPI_SIGMOID) x = sigmoid_fast(x);
                            op[i] = x;
                        }
                        if (valid[g]) *(float4*)(a.YF + (size_t)row[g] * a.ldYF + f0) = make_float4(op[0], op[1], op[2], op[3]);
                    }
                }
            }
        } else if (EPI == EPI_BERN) {
            // log p(x|z) = sum_j x_j l_j - softplus(l_j), softplus(l) = max(l,0) + ln2*log2(1 + 2^(-|l| log2e))  (iwae1.py:111)
            // wave-uniform branch: only the last pixel group needs masks
            // keep: the training step also stores s = x - sigmoid(l) (bf16, P-layout) -- d lpxz / d l up to the row weight,
            // which out_bwd and the output layer's weight gradient then read instead of recomputing the logits.
            auto bern_body = [&](auto masked, auto keep) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        f32x4 t0 = acc[2 * p][g], t1 = acc[2 * p + 1][g];
                        if (!kBiasInit) {
                            t0 += (f32x4){bias_of(2 * p, 0), bias_of(2 * p, 1), bias_of(2 * p, 2), bias_of(2 * p, 3)};
                            t1 += (f32x4){bias_of(2 * p + 1, 0), bias_of(2 * p + 1, 1), bias_of(2 * p + 1, 2), bias_of(2 * p + 1, 3)};
                        }
                        rowacc[g] += bern8<decltype(masked)::value, decltype(keep)::value>(t0, t1, pre[p][g], 64 * mg + 32 * p + 4 * q, a.Xdim, stP[p][g]);
                    }
            };
            const bool full = 64 * mg + 64 <= a.Xdim;
            if (a.YP) {
                if (full) bern_body(std::false_type{}, std::true_type{});
                else bern_body(std::true_type{}, std::true_type{});
                st_mg = mg;
            } else {
                if (full) bern_body(std::false_type{}, std::false_type{});
                else bern_body(std::true_type{}, std::false_type{});
            }
            if (a.logits_out) {     // rare path (the reference dict's "logits"): reference [k,B,X] order
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int f = 64 * mg + 32 * p + 16 * (j >> 2) + 4 * q + (j & 3);
                            if (valid[g] && f < a.Xdim)
                                a.logits_out[((size_t)sidx[g] * a.B + bidx[g]) * a.Xdim + f] = acc[2 * p + (j >> 2)][g][j & 3] + (kBiasInit ? 0.0f : bias_of(2 * p + (j >> 2), j & 3));
                        }
            }
        }
        if (kPre) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int g = 0; g < G; ++g) pre[p][g] = pre_n[p][g];
        }
        DS_STAMP(5)      // epilogue math
    }
    emit_stores();

    if (EPI == EPI_BERN) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float v = rowacc[g];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0 && valid[g]) a.lpxz[(size_t)blockIdx.y * a.lpxz_stride + row[g]] = v;      // stride 0: one block owns all pixel groups
        }
    }
#ifdef IWAE_DENSE_STAMPS
    DS_STAMP(6)      // tail
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWV + wave) * 8 + i] = ds_sum[i];
    }
#endif
}

// ---------------------------------------------------------------------------------
// block_fwd_kernel: a whole BasicBlock (iwae1.py:36-44: two tanh layers + the mu | sigma head) on a SMALL number of rows --
// the encoder on the B images of a batch -- in ONE launch.  As three dense_kernel launches this is a chain of
// latency-bound kernels (B = 1024: 15 + 8 + 5 us on 32 workgroups each); here one 16-wave workgroup owns 16 rows through
// all three layers:
//   * wave w owns out-feature tile w (16 features) of every layer (hidden <= 256, head <= 2 x 128: at most 16 tiles);
//   * weights never touch LDS: each wave reads ITS tile's A fragments straight from the L2-resident image (1 KiB per
//     wave instruction, coalesced), AD fragments in flight; the second layer's fragments are requested before the first
//     layer starts and the head's while it finishes.  <= 96 registers: a workgroup (16 waves) leaves a quarter of the CU's
//     register file to whatever else is resident (the deferred decoder update runs beside the encoder);
//   * activations (the shared B operand) live in LDS as 1 KiB k-step blocks in the A-image's conflict-free arrangement;
//     a layer's output goes back there (and to HBM for the backward pass) as bf16, 8 bytes per lane.
// ---------------------------------------------------------------------------------
#define BLOCKFWD_MAX_KT 8      // k-steps of the two later layers (hidden <= 256)
template <int AD, bool OUT = false>
__global__ __launch_bounds__(1024, ((AD > 6 || OUT) ? 2 : 5)) void block_fwd_kernel(BlockFwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * 16;
    const int row = r0 + rho;
    const bool valid = row < a.R;
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    char* act0 = smem;                                   // x tile: KT0 blocks
    char* act1 = smem + (size_t)a.KT0 * 1024;            // h1: KT1 blocks
    char* act2 = act1 + (size_t)a.KT1 * 1024;            // h2: KT1 blocks

    // stage the x tile (rows beyond R: zeros).  Xf != null: the rows arrive as fp32 [R][Xdim] (the batch as the caller
    // handed it over) and are converted here -- prep_rows_kernel's job, without its launch; the bf16 P-layout rows also go
    // to HBM (XPout) for the kernels that read x later (Bernoulli forward, weight gradient of the first layer).
    // sample != 0: the rows are z = mu + sigma*eps, made here in sample_kernel's arithmetic and summation order: wave t makes the
    // 32-feature step t of the 16 rows (all loads of the tile in flight at once; sample_kernel walks its steps in turn), the lanes'
    // partial log-density sums meet in LDS and wave 15 adds them up once the tile is staged.
    float* red = (float*)(act2 + (size_t)a.KT1 * 1024);      // [3][KT0][64]
    if (a.sample) {
        if (wave < a.KT0) {
            const int rowc = valid ? row : a.S.M - 1;
            const int b = rowc / a.S.k, sidx = rowc - b * a.S.k;
            const float* hd = a.S.head + (size_t)(a.S.head_per_row ? rowc : b) * a.S.ldH;
            float z8[8], lp = 0.0f, lq = 0.0f, lq2 = 0.0f;
            sample_step(a.S, wave, q, rowc, b, sidx, hd, z8, lp, lq, lq2);
            const uint4 v = valid ? make_uint4(pack2(z8[0], z8[1]), pack2(z8[2], z8[3]), pack2(z8[4], z8[5]), pack2(z8[6], z8[7])) : make_uint4(0, 0, 0, 0);
            *(uint4*)(act0 + wave * 1024 + a_off) = v;
            if (valid) *(uint4*)(a.S.ZP + (size_t)row * a.S.Dp + 8 * (4 * wave + q)) = v;
            red[(0 * a.KT0 + wave) * 64 + lane] = lp;
            red[(1 * a.KT0 + wave) * 64 + lane] = lq;
            red[(2 * a.KT0 + wave) * 64 + lane] = lq2;
        }
    } else
    for (int c = threadIdx.x; c < a.KT0 * 64; c += 1024) {      // 16-byte chunks: (ks, row, quad)
        const int ks = c >> 6, rr = (c >> 2) & 15, qq = c & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + rr < a.R) {
            if (a.Xf) {
                float t[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int f0 = 32 * ks + 16 * h + 4 * qq;
                    const float* src = a.Xf + (size_t)(r0 + rr) * a.Xdim + f0;
                    if ((a.Xdim & 3) == 0 && f0 + 3 < a.Xdim) {
                        const float4 x4 = *(const float4*)src;
                        t[4 * h] = x4.x; t[4 * h + 1] = x4.y; t[4 * h + 2] = x4.z; t[4 * h + 3] = x4.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[4 * h + i] = (f0 + i < a.Xdim) ? src[i] : 0.0f;
                    }
                }
                v = make_uint4(pack2(t[0], t[1]), pack2(t[2], t[3]), pack2(t[4], t[5]), pack2(t[6], t[7]));
                *(uint4*)(a.XPout + (size_t)(r0 + rr) * a.ldX + ks * 32 + qq * 8) = v;
            } else {
                v = *(const uint4*)(a.X + (size_t)(r0 + rr) * a.ldX + ks * 32 + qq * 8);
            }
        }
        *(uint4*)(act0 + ks * 1024 + rr * 64 + ((qq ^ hperm(rr >> 2)) * 16)) = v;
    }
    // the later layers' weights: requested now, used after the barriers
    const int mg = wave >> 2, tg = wave & 3;
    const bool has1 = wave < a.NT1, has2 = wave < a.NT2;
    const char* w0 = a.img0 + (size_t)mg * img_mg_group_bytes(a.KT0) + tg * 1024 + a_off;
    const char* w1 = a.img1 + (size_t)mg * img_mg_group_bytes(a.KT1) + tg * 1024 + a_off;
    const char* w2 = a.img2 + (size_t)mg * img_mg_group_bytes(a.KT1) + tg * 1024 + a_off;
    uint4 A1[BLOCKFWD_MAX_KT], A2[BLOCKFWD_MAX_KT];
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks) {
        A1[ks] = make_uint4(0, 0, 0, 0);
        if (ks < a.KT1 && has1) A1[ks] = *(const uint4*)(w1 + (size_t)ks * 4096);
    }
    const float4 b0 = has1 ? *(const float4*)(a.img0 + (size_t)mg * img_mg_group_bytes(a.KT0) + (size_t)a.KT0 * 4096 + (16 * tg + 4 * q) * 4) : make_float4(0, 0, 0, 0);
    const float4 b1 = has1 ? *(const float4*)(a.img1 + (size_t)mg * img_mg_group_bytes(a.KT1) + (size_t)a.KT1 * 4096 + (16 * tg + 4 * q) * 4) : make_float4(0, 0, 0, 0);
    const float4 b2 = has2 ? *(const float4*)(a.img2 + (size_t)mg * img_mg_group_bytes(a.KT1) + (size_t)a.KT1 * 4096 + (16 * tg + 4 * q) * 4) : make_float4(0, 0, 0, 0);

    // out-feature tile `wave` of a tanh layer -> bf16, 8 bytes per lane: half a k-step fragment of the next layer's operand
    auto emit_tanh = [&](const f32x4& acc, const float4& b, char* act, uint16_t* YP, int ldY) {
        const uint2 v = make_uint2(pack2(tanh_fast(acc[0] + b.x), tanh_fast(acc[1] + b.y)), pack2(tanh_fast(acc[2] + b.z), tanh_fast(acc[3] + b.w)));
        const int ks = wave >> 1, h = wave & 1;
        *(uint2*)(act + ks * 1024 + a_off + 8 * h) = v;
        if (valid) *(uint2*)(YP + (size_t)row * ldY + ks * 32 + q * 8 + 4 * h) = v;
    };

    // ---- layer 1: K = KT0 k-steps, A fragments streamed AD deep
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        uint4 av[AD];
#pragma unroll
        for (int i = 0; i < AD; ++i) av[i] = (i < a.KT0 && has1) ? *(const uint4*)(w0 + (size_t)i * 4096) : make_uint4(0, 0, 0, 0);
        __syncthreads();                         // x tile staged
        for (int k0 = 0; k0 < a.KT0; k0 += AD) {
#pragma unroll
            for (int i = 0; i < AD; ++i) {
                const int ks = k0 + i;
                if (ks < a.KT0) {
                    const uint4 bv = *(const uint4*)(act0 + ks * 1024 + a_off);
                    acc = mfma16(av[i], bv, acc);
                    if (ks + AD < a.KT0 && has1) av[i] = *(const uint4*)(w0 + (size_t)(ks + AD) * 4096);
                }
            }
        }
    }
    // the head's weights: requested here (the first layer's stream registers are free), used two barriers later
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks) {
        A2[ks] = make_uint4(0, 0, 0, 0);
        if (ks < a.KT1 && has2) A2[ks] = *(const uint4*)(w2 + (size_t)ks * 4096);
    }
    if (a.sample && wave == 15) {      // the rows' log-densities: the lane's steps in turn, then the four quads (sample_kernel's order)
        float lp = 0.0f, lq = 0.0f, lq2 = 0.0f;
        for (int t = 0; t < a.KT0; ++t) {
            lp += red[(0 * a.KT0 + t) * 64 + lane];
            lq += red[(1 * a.KT0 + t) * 64 + lane];
            lq2 += red[(2 * a.KT0 + t) * 64 + lane];
        }
        lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
        lq += __shfl_xor(lq, 16); lq += __shfl_xor(lq, 32);
        lq2 += __shfl_xor(lq2, 16); lq2 += __shfl_xor(lq2, 32);
        if (q == 0 && valid) {
            if (a.S.lp_prior) a.S.lp_prior[row] = lp;
            a.S.lq[row] = lq;
            if (a.S.lq_dreg) a.S.lq_dreg[row] = lq2;
        }
    }
    if (has1) emit_tanh(acc, b0, act1, a.H1, a.ldH);
    __syncthreads();
    // ---- layer 2
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
        if (ks < a.KT1) acc = mfma16(A1[ks], *(const uint4*)(act1 + ks * 1024 + a_off), acc);
    if (has1) emit_tanh(acc, b1, act2, a.H2, a.ldH);
    __syncthreads();
    // ---- head: mu | sigma = exp(.) + 1e-6 (iwae1.py:34,42), fp32 natural order
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
        if (ks < a.KT1) acc = mfma16(A2[ks], *(const uint4*)(act2 + ks * 1024 + a_off), acc);
    if (has2 && valid) {
        const int f0 = 16 * wave + 4 * q;
        float o[4] = {acc[0] + b2.x, acc[1] + b2.y, acc[2] + b2.z, acc[3] + b2.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (f0 + i >= a.split) o[i] = exp2_raw(o[i] * LOG2E_F) + 1e-6f;
        *(float4*)(a.YF + (size_t)row * a.ldYF + f0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if constexpr (OUT) {
        // ---- the Bernoulli output layer of the decoder on the tile's 16 rows (iwae1.py:83,111): wave w takes the 32-pixel halves w, w + 16 of the
        // pixel range -- two accumulator tiles each, weight fragments straight from the L2-resident image, g2 from LDS (act2, complete behind
        // the barrier above) -- and bern8's epilogue; the rows' sums over the waves meet in LDS in wave order.
        float rowacc = 0.0f;
        const int bimg = min(row, a.R - 1) / a.ok;
        for (int h = wave; h < a.oH; h += 16) {
            const int mg = h >> 1, tg = 2 * (h & 1);
            const char* wb = a.oimg + (size_t)mg * img_mg_group_bytes(a.KT1) + a_off;
            uint4 A0[BLOCKFWD_MAX_KT], A1v[BLOCKFWD_MAX_KT];
#pragma unroll
            for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks) {
                A0[ks] = make_uint4(0, 0, 0, 0); A1v[ks] = make_uint4(0, 0, 0, 0);
                if (ks < a.KT1) { A0[ks] = *(const uint4*)(wb + (size_t)(ks * 4 + tg) * 1024); A1v[ks] = *(const uint4*)(wb + (size_t)(ks * 4 + tg + 1) * 1024); }
            }
            const float* bb = (const float*)(a.oimg + (size_t)mg * img_mg_group_bytes(a.KT1) + (size_t)a.KT1 * 4096) + 16 * tg + 4 * q;
            const float4 c0 = *(const float4*)bb, c1 = *(const float4*)(bb + 16);
            const uint4 xb = *(const uint4*)(a.oXB + (size_t)bimg * a.oldXB + 32 * h + 8 * q);
            f32x4 l0 = (f32x4){c0.x, c0.y, c0.z, c0.w}, l1 = (f32x4){c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
                if (ks < a.KT1) {
                    const uint4 bv = *(const uint4*)(act2 + ks * 1024 + a_off);
                    l0 = mfma16(A0[ks], bv, l0);
                    l1 = mfma16(A1v[ks], bv, l1);
                }
            uint4 sp = make_uint4(0, 0, 0, 0);
            const bool full = 32 * h + 32 <= a.oXdim;
            if (a.oSP) rowacc += full ? bern8<false, true>(l0, l1, xb, 32 * h + 4 * q, a.oXdim, sp) : bern8<true, true>(l0, l1, xb, 32 * h + 4 * q, a.oXdim, sp);
            else rowacc += full ? bern8<false, false>(l0, l1, xb, 32 * h + 4 * q, a.oXdim, sp) : bern8<true, false>(l0, l1, xb, 32 * h + 4 * q, a.oXdim, sp);
            if (a.oSP && valid) *(uint4*)(a.oSP + (size_t)row * a.oldS + 32 * h + 8 * q) = sp;
        }
        rowacc += __shfl_xor(rowacc, 16);
        rowacc += __shfl_xor(rowacc, 32);
        float* red2 = (float*)act0;        // (the input tile is no longer read: every wave is past the second barrier)
        if (q == 0) red2[wave * 16 + rho] = rowacc;
        __syncthreads();
        if (wave == 0 && q == 0 && valid) {
            float v = 0.0f;
            for (int w = 0; w < 16; ++w) v += red2[w * 16 + rho];
            a.olpxz[row] = v;
        }
    }
}

// ---------------------------------------------------------------------------------
// block_bwd_kernel: the dX chain of a BasicBlock on few rows (the encoder's backward pass on the B images) in ONE launch, the
// mirror image of block_fwd_kernel: d2 = (dhead Whead^T) * (1 - h2^2), d1 = (d2 W2^T) * (1 - h1^2).  A 16-wave workgroup owns 16
// rows, wave w owns hidden tile w of both products, the backward images' fragments come straight from L2 into registers (the
// second layer's are requested before the first product starts), the data operands sit in LDS as 1 KiB k-step blocks.
// As two dense_kernel<EPI_DX> launches this was 6.5 + 6.4 us alone and 20 - 35 us beside the weight gradients.
// ---------------------------------------------------------------------------------
// the per-image transforms behind the sums over the samples (latent_bwd_kernel's last step): the KL terms of vae_elbo_kl, d sigma -> d pre-activation of the exp
__device__ __forceinline__ void latent_finish(const LatentBwdArgs& a, const int f0, const float (&mu)[4], const float (&sgm)[4], float (&dmu)[4], float (&dsg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (f0 + i < a.D) {
            dmu[i] += a.kmu * mu[i];
            dsg[i] += a.ksig * (sgm[i] - 1.0f / sgm[i]);
            dsg[i] *= (sgm[i] - 1e-6f);
        } else { dmu[i] = 0.0f; dsg[i] = 0.0f; }
    }
}
// latent_bwd_kernel's sums for ONE image by ONE wave (block_bwd_kernel with lat_on: the image encoder's backward on few rows): lanes = 32 feature
// quads x 2 sample groups, four samples' loads in flight per lane; on return lanes < 32 hold d mu and d sigma -> pre-activation of the exp
// for features 4 lane .. 4 lane + 3 (zeros beyond D).  The same arithmetic as latent_bwd_kernel, another summation order; no
// conditional prior (the caller keeps the separate kernel for that).
// part / nparts: this wave's share of the image's samples (s = sg + 2 part + 2 nparts j); FINISH: the whole image in this wave (nparts = 1):
// the cross-half add and the final transforms follow here, else the caller combines the parts' raw sums (lanes < 32 after the shuffle) first
template <bool FINISH>
__device__ __forceinline__ void latent_image_part(const LatentBwdArgs& a, const int b, const int lane, const int part, const int nparts, float (&dmu)[4], float (&dsg)[4],
                                                  float (&mu)[4], float (&sgm)[4]) {
    const int f4 = lane & 31, sg = lane >> 5, f0 = 4 * f4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { dmu[i] = 0.0f; dsg[i] = 0.0f; mu[i] = 0.0f; sgm[i] = 1.0f; }
    if (f0 < a.D) {
        float rs2[4], rsg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = f0 + i < a.D;
            mu[i] = ok ? a.head[(size_t)b * a.ldH + f0 + i] : 0.0f;
            sgm[i] = ok ? a.head[(size_t)b * a.ldH + a.Dp + f0 + i] : 1.0f;
            const float s2 = sgm[i] + 1e-6f;
            rs2[i] = 1.0f / (s2 * s2);
            rsg[i] = 1.0f / sgm[i];
        }
        constexpr int UN = 4;
        const int sstep = 2 * nparts;
        for (int s0 = sg + 2 * part; s0 < a.k; s0 += sstep * UN) {
            float4 dz[UN], cf[UN];
            float e[UN][4];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int s = s0 + sstep * u, sc = s < a.k ? s : a.k - 1, row = b * a.k + sc;      // clamped, weighted by 0 below
                if (a.dzh) { const uint2 h2 = *(const uint2*)(a.dzh + (size_t)row * a.ldDZ + f0); dz[u] = make_float4(bflo(h2.x), bfhi(h2.x), bflo(h2.y), bfhi(h2.y)); }
                else dz[u] = *(const float4*)(a.dz + (size_t)row * a.ldDZ + f0);
                if (a.dz2) {
                    const float4 t2 = *(const float4*)(a.dz2 + (size_t)row * a.ldDZ + f0), t3 = *(const float4*)(a.dz3 + (size_t)row * a.ldDZ + f0);
                    dz[u] = make_float4(dz[u].x + t2.x + t3.x, dz[u].y + t2.y + t3.y, dz[u].z + t2.z + t3.z, dz[u].w + t2.w + t3.w);
                }
                cf[u] = a.cf[row];
                eps4(a.eps, b, sc, row, f4, a.D, e[u]);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (s0 + sstep * u >= a.k) cf[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                const float dzv[4] = {dz[u].x, dz[u].y, dz[u].z, dz[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (f0 + i < a.D) {
                        const float z = mu[i] + sgm[i] * e[u][i];
                        const float t = cf[u].x * dzv[i] + cf[u].y * z + cf[u].z * (z - mu[i]) * rs2[i];      // (N(0,1) prior: up = z, 1/sigma_p = 1)
                        dmu[i] += t;
                        dsg[i] += t * e[u][i] + cf[u].w * rsg[i];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { dmu[i] += __shfl_xor(dmu[i], 32); dsg[i] += __shfl_xor(dsg[i], 32); }
    if constexpr (FINISH) latent_finish(a, f0, mu, sgm, dmu, dsg);
}
__device__ __forceinline__ void latent_image_wave(const LatentBwdArgs& a, const int b, const int lane, float (&dmu)[4], float (&dsg)[4]) {
    float mu[4], sgm[4];
    latent_image_part<true>(a, b, lane, 0, 1, dmu, dsg, mu, sgm);
}
// RW = data rows per workgroup.  16: the tile is full.  4 (round 5, lat_on with many samples per image): a QUARTER-filled tile per workgroup and four
// times the workgroups -- the latent sums in front of the dX chain read k x (dz + draws) per image, which 64 workgroups (a wave per image) pull at a
// quarter of the chip's bandwidth (k = 50: +10 us over the separate latent_bwd_kernel, round 4); with 4 images per workgroup the 16 waves take
// an image's samples four ways (partial sums meet in LDS in part order) on 256 workgroups, and latent_bwd_kernel's launch is gone from the step's
// main chain at k = 50 too (13.8 + 8.6 us alone + a boundary -> one launch).  The 12 idle rows of the MFMA tile cost nothing that matters (0.16 GFLOP).
template <int RW>
__global__ __launch_bounds__(1024, 4) void block_bwd_kernel(BlockBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * RW;
    const int row = r0 + rho;
    const bool valid = rho < RW && row < a.R;
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    char* act0 = smem;                                   // dhead tile: KTH blocks
    char* act1 = smem + (size_t)a.KTH * 1024;            // d2 tile: KT1 blocks
    if (RW == 4 && a.lat_on) {
        // wave w: image r0 + (w >> 2), part w & 3 of its samples; raw sums -> LDS; waves 0, 4, 8, 12 add the four parts in part order, finish, and
        // write their image's row of the dhead tile (rows 4..15 of the tile: zeros)
        float* part_lds = (float*)(smem + (size_t)(a.KTH + a.KT1) * 1024);      // [4 images][4 parts][32 lanes][8]
        for (int c = threadIdx.x; c < a.KTH * 64; c += 1024) *(uint4*)(act0 + c * 16) = make_uint4(0, 0, 0, 0);
        float dmu[4], dsg[4], mu[4], sgm[4];
        const int img = wave >> 2, part = wave & 3, b = r0 + img, Dp = a.lat.Dp;
        if (b < a.R) latent_image_part<false>(a.lat, b, lane, part, 4, dmu, dsg, mu, sgm);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { dmu[i] = 0.0f; dsg[i] = 0.0f; mu[i] = 0.0f; sgm[i] = 1.0f; }
        }
        if (lane < 32) {
            float* pl = part_lds + ((size_t)(img * 4 + part) * 32 + lane) * 8;
            *(float4*)pl = make_float4(dmu[0], dmu[1], dmu[2], dmu[3]);
            *(float4*)(pl + 4) = make_float4(dsg[0], dsg[1], dsg[2], dsg[3]);
        }
        __syncthreads();      // (also: the zero fill of the tile is complete)
        if (part == 0 && lane < 32) {
#pragma unroll
            for (int pp = 1; pp < 4; ++pp) {
                const float* pl = part_lds + ((size_t)(img * 4 + pp) * 32 + lane) * 8;
                const float4 m4 = *(const float4*)pl, s4 = *(const float4*)(pl + 4);
                dmu[0] += m4.x; dmu[1] += m4.y; dmu[2] += m4.z; dmu[3] += m4.w;
                dsg[0] += s4.x; dsg[1] += s4.y; dsg[2] += s4.z; dsg[3] += s4.w;
            }
            latent_finish(a.lat, 4 * lane, mu, sgm, dmu, dsg);
            if (lane < Dp / 4) {
                const int f0 = 4 * lane;
                const uint2 vm = make_uint2(pack2(dmu[0], dmu[1]), pack2(dmu[2], dmu[3])), vs = make_uint2(pack2(dsg[0], dsg[1]), pack2(dsg[2], dsg[3]));
                const int pm = p_pos(f0), ps = p_pos(Dp + f0);
                *(uint2*)(act0 + (pm >> 5) * 1024 + img * 64 + ((((pm & 31) >> 3) ^ hperm(img >> 2)) * 16) + ((pm >> 2) & 1) * 8) = vm;
                *(uint2*)(act0 + (ps >> 5) * 1024 + img * 64 + ((((ps & 31) >> 3) ^ hperm(img >> 2)) * 16) + ((ps >> 2) & 1) * 8) = vs;
                if (b < a.R && a.lat.DHP) {
                    *(uint2*)(a.lat.DHP + (size_t)b * (2 * Dp) + pm) = vm;
                    *(uint2*)(a.lat.DHP + (size_t)b * (2 * Dp) + ps) = vs;
                }
            }
        }
    } else
    if (a.lat_on) {      // the dhead tile is MADE here: wave w = image r0 + w (latent_bwd_kernel's job, without its launch)
        float dmu[4], dsg[4];
        const int b = r0 + wave, Dp = a.lat.Dp;
        if (b < a.R) latent_image_wave(a.lat, b, lane, dmu, dsg);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { dmu[i] = 0.0f; dsg[i] = 0.0f; }
        }
        if (lane < Dp / 4 && wave < 16) {
            const int f0 = 4 * lane;
            const uint2 vm = make_uint2(pack2(dmu[0], dmu[1]), pack2(dmu[2], dmu[3])), vs = make_uint2(pack2(dsg[0], dsg[1]), pack2(dsg[2], dsg[3]));
            const int pm = p_pos(f0), ps = p_pos(Dp + f0);
            *(uint2*)(act0 + (pm >> 5) * 1024 + wave * 64 + ((((pm & 31) >> 3) ^ hperm(wave >> 2)) * 16) + ((pm >> 2) & 1) * 8) = vm;
            *(uint2*)(act0 + (ps >> 5) * 1024 + wave * 64 + ((((ps & 31) >> 3) ^ hperm(wave >> 2)) * 16) + ((ps >> 2) & 1) * 8) = vs;
            if (b < a.R && a.lat.DHP) {
                *(uint2*)(a.lat.DHP + (size_t)b * (2 * Dp) + pm) = vm;
                *(uint2*)(a.lat.DHP + (size_t)b * (2 * Dp) + ps) = vs;
            }
        }
    } else
    for (int c = threadIdx.x; c < a.KTH * 64; c += 1024) {      // 16-byte chunks: (ks, row, quad)
        const int ks = c >> 6, rr = (c >> 2) & 15, qq = c & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (rr < RW && r0 + rr < a.R) v = *(const uint4*)(a.DH + (size_t)(r0 + rr) * a.ldDH + ks * 32 + qq * 8);
        *(uint4*)(act0 + ks * 1024 + rr * 64 + ((qq ^ hperm(rr >> 2)) * 16)) = v;
    }
    const int mg = wave >> 2, tg = wave & 3;
    const bool has = wave < a.NT1;
    const char* wh = a.imgH + (size_t)mg * img_mg_group_bytes(a.KTH) + tg * 1024 + a_off;
    const char* w2 = a.imgL2 + (size_t)mg * img_mg_group_bytes(a.KT1) + tg * 1024 + a_off;
    uint4 AH[BLOCKFWD_MAX_KT], A2[BLOCKFWD_MAX_KT];
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks) {
        AH[ks] = make_uint4(0, 0, 0, 0); A2[ks] = make_uint4(0, 0, 0, 0);
        if (ks < a.KTH && has) AH[ks] = *(const uint4*)(wh + (size_t)ks * 4096);
        if (ks < a.KT1 && has) A2[ks] = *(const uint4*)(w2 + (size_t)ks * 4096);
    }
    // the lane's 4 activations of hidden tile `wave` (k-step wave >> 1, half wave & 1 of the P-layout row)
    const int kso = wave >> 1, hh = wave & 1;
    uint2 y2 = make_uint2(0, 0), y1 = make_uint2(0, 0);
    if (valid && has) {
        y2 = *(const uint2*)(a.H2 + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh);
        y1 = *(const uint2*)(a.H1 + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh);
    }
    auto dtanh = [&](const f32x4& acc, const uint2& y) {
        const float ya = bflo(y.x), yb = bfhi(y.x), yc = bflo(y.y), yd = bfhi(y.y);
        return make_uint2(pack2(acc[0] * (1.0f - ya * ya), acc[1] * (1.0f - yb * yb)), pack2(acc[2] * (1.0f - yc * yc), acc[3] * (1.0f - yd * yd)));
    };
    __syncthreads();
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
        if (ks < a.KTH) acc = mfma16(AH[ks], *(const uint4*)(act0 + ks * 1024 + a_off), acc);
    if (has) {
        const uint2 v = valid ? dtanh(acc, y2) : make_uint2(0, 0);
        *(uint2*)(act1 + kso * 1024 + a_off + 8 * hh) = v;
        if (valid) *(uint2*)(a.D2 + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh) = v;
    }
    __syncthreads();
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
        if (ks < a.KT1) acc = mfma16(A2[ks], *(const uint4*)(act1 + ks * 1024 + a_off), acc);
    if (has && valid) *(uint2*)(a.D1 + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh) = dtanh(acc, y1);
}


// Wave-wide sum / max through the DPP path (row_shr 1, 2, 4, 8, then row_bcast 15 and 31: the total arrives in lane 63 and is handed to
// every lane through an SGPR): six vector instructions of ~10 cycles each, where the ds_bpermute butterfly of __shfl_xor is six LDS
// round trips of ~100 -- lse_image is four such reductions deep, on the tail of the decoder kernel.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_or(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_or<0x111, 0xf>(0.0f, v); v += dpp_or<0x112, 0xf>(0.0f, v); v += dpp_or<0x114, 0xf>(0.0f, v); v += dpp_or<0x118, 0xf>(0.0f, v);
    v += dpp_or<0x142, 0xa>(0.0f, v);
    v += dpp_or<0x143, 0xc>(0.0f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_or<0x111, 0xf>(-INFINITY, v)); v = fmaxf(v, dpp_or<0x112, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_or<0x114, 0xf>(-INFINITY, v)); v = fmaxf(v, dpp_or<0x118, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_or<0x142, 0xa>(-INFINITY, v));
    v = fmaxf(v, dpp_or<0x143, 0xc>(-INFINITY, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// What one wave does for image b: log_w over its k samples, logmeanexp / softmax / objective gradients (iwae1.py:113-139).  `src` says where
// the rows' terms come from: LseGlobalSrc (lse_kernel: the arrays of LseArgs) or the LDS copies bern_pipe_kernel keeps of what it made itself.
struct LseGlobalSrc {
    __device__ __forceinline__ float px(const LseArgs& a, int, int r) const {
        float px = a.term[0][r];
        if (a.n_px_part > 1) {      // log p(x|z) as partial sums over pixel groups (small row counts): fixed order
            // (all partials requested at once: as a loop of dependent adds each load waited for the one before it -- 13 L2 round
            // trips, most of this kernel's 10 us at B = 20)
            float part[16];
#pragma unroll
            for (int i = 1; i < 16; ++i) part[i] = (i < a.n_px_part) ? a.term[0][(size_t)i * a.px_stride + r] : 0.0f;
#pragma unroll
            for (int i = 1; i < 16; ++i) px += part[i];
            for (int i = 16; i < a.n_px_part; ++i) px += a.term[0][(size_t)i * a.px_stride + r];
            a.term0_out[r] = px;
        }
        return px;
    }
    __device__ __forceinline__ float px_total(const LseArgs& a, int, int r) const { return a.term0_out[r]; }      // (total log p(x|z), written by px() above)
    __device__ __forceinline__ float term(const LseArgs& a, int t, int, int r) const { return a.term[t][r]; }
    __device__ __forceinline__ float lq_dreg(const LseArgs& a, int, int r) const { return a.lq_dreg[r]; }
};

// RED = the team working on one image: a wave (WaveRed: the training step's k) or a whole workgroup (BlockRed<NW>: the k = 5000 evaluator, where a wave
// per image is ~100 waves on the machine walking 5 000 samples each: 26 % of the bf16 evaluator's time in round 3's profile).  `lane` = index in the team.
struct WaveRed {
    static constexpr int NT = 64;
    __device__ __forceinline__ float sum(float v) const { return wave_sum(v); }
    __device__ __forceinline__ float max(float v) const { return wave_max(v); }
};
template <int NW>
struct BlockRed {      // (every thread of the workgroup calls sum / max the same number of times: they hold barriers)
    static constexpr int NT = 64 * NW;
    float* slots;      // NW floats of LDS
    __device__ __forceinline__ float sum(float v) const {
        const float w = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = w;
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < NW; ++i) t += slots[i];
        return t;
    }
    __device__ __forceinline__ float max(float v) const {
        const float w = wave_max(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = w;
        __syncthreads();
        float t = -INFINITY;
#pragma unroll
        for (int i = 0; i < NW; ++i) t = fmaxf(t, slots[i]);
        return t;
    }
};

template <class SRC, class RED = WaveRed>
__device__ __forceinline__ void lse_image(const LseArgs& a, const int b, const int lane, const SRC& src, const RED red = RED()) {
    constexpr int NT = RED::NT;
    const int k = a.k;
    const bool single = NT == 64 && k <= 64;          // one sample per lane: log_w stays in a register between the passes
    // the image's head (for the KL term at the end): requested first, so that its round trip runs beside the log_w terms' instead of
    // behind the whole kernel
    float kmu[2] = {0.0f, 0.0f}, ksg[2] = {1.0f, 1.0f};
    if (a.head) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = lane + NT * i;
            if (f < a.D) { kmu[i] = a.head[(size_t)b * a.ldH + f]; ksg[i] = a.head[(size_t)b * a.ldH + a.Dp + f]; }
        }
    }
    float lw_reg = 0.0f;
    float m = -INFINITY, sum_lw = 0.0f, sum_px = 0.0f, sum_t1 = 0.0f, sum_t2 = 0.0f;
    for (int s = lane; s < k; s += NT) {
        const int r = b * k + s;
        const float px = src.px(a, s, r);
        float lw = a.coef[0] * px;
#pragma unroll
        for (int t = 1; t < 5; ++t)
            if (a.term[t]) lw += a.coef[t] * src.term(a, t, s, r);
        a.logw[r] = lw;
        lw_reg = lw;
        m = fmaxf(m, lw);
        sum_lw += lw;
        sum_px += px;
        if (a.term[1]) sum_t1 += src.term(a, 1, s, r);
        if (a.term[2]) sum_t2 += src.term(a, 2, s, r);
    }
    m = red.max(m); sum_lw = red.sum(sum_lw); sum_px = red.sum(sum_px); sum_t1 = red.sum(sum_t1); sum_t2 = red.sum(sum_t2);
    float se = 0.0f;
    for (int s = lane; s < k; s += NT) se += __expf((single ? lw_reg : a.logw[b * k + s]) - m);
    se = red.sum(se);
    const float inv_se = 1.0f / se;
    const float invB = 1.0f / (float)a.B, invkB = invB / (float)k;
    float eq14 = 0.0f, dreg = 0.0f;
    if (!a.lme_only)
    for (int s = lane; s < k; s += NT) {
        const int r = b * k + s;
        const float lw = single ? lw_reg : a.logw[r];
        const float wn = __expf(lw - m) * inv_se;       // iwae1.py:128-131 == softmax over k (:137)
        eq14 += wn * lw;
        a.wn[r] = wn;
        float G;                                         // dLoss/dlog_w, loss = -objective (iwae1.py:157)
        float4 cf = make_float4(1.0f, 0.0f, 0.0f, 0.0f); // (ca, cz, cq, cs) for latent_bwd_kernel
        if (a.objective == OBJ_VAE_ELBO) {
            G = -invkB; cf.y = -G * a.beta * a.cz_on; cf.w = G * a.beta;
        } else if (a.objective == OBJ_VAE_ELBO_KL) {
            G = -invkB;
        } else if (a.objective == OBJ_DREG) {            // tasks/task02.py:61-76,95-96
            G = -wn * invB;
            const float c2 = wn * wn * invB;
            cf.x = wn; cf.y = c2; cf.z = -c2;
        } else {                                         // iwae_elbo, iwae_eq14 (same gradient)
            G = -wn * invB; cf.y = -G * a.beta * a.cz_on; cf.w = G * a.beta;
        }
        if (a.lq_dreg) dreg += wn * wn * (src.term(a, 1, s, r) + src.px_total(a, s, r) - src.lq_dreg(a, s, r));   // tasks/task02.py:70-73
        a.gx[r] = G;
        if (a.gx_local && r >= a.gx_r0 && r < a.gx_r0 + a.gx_n) a.gx_local[r - a.gx_r0] = G;
        a.cf[r] = cf;
    }
    eq14 = red.sum(eq14); dreg = red.sum(dreg);
    // KL(q(z|x) || N(0,1)) per image (iwae1.py:116), TFP closed form
    float kl = 0.0f;
    if (a.head) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (lane + NT * i < a.D) {
                const float ls = __logf(ksg[i]);
                kl += 0.5f * kmu[i] * kmu[i] + 0.5f * expm1f(2.0f * ls) - ls;
            }
        }
        for (int f = lane + 2 * NT; f < a.D; f += NT) {      // (latent widths beyond 128)
            const float mu = a.head[(size_t)b * a.ldH + f], sg = a.head[(size_t)b * a.ldH + a.Dp + f];
            const float ls = __logf(sg);
            kl += 0.5f * mu * mu + 0.5f * expm1f(2.0f * ls) - ls;
        }
        kl = red.sum(kl);
    }
    if (lane == 0) {
        float* pb = a.per_b;
        const int B = a.B;
        pb[PB_LME * B + b] = m + __logf(se / (float)k);          // utils.py:6-8
        pb[PB_MEAN * B + b] = sum_lw / (float)k;
        pb[PB_EQ14 * B + b] = eq14;
        pb[PB_KL * B + b] = kl;
        pb[PB_PX * B + b] = sum_px / (float)k;
        pb[PB_T1 * B + b] = sum_t1 / (float)k;
        pb[PB_T2 * B + b] = sum_t2 / (float)k;
        pb[PB_DREG * B + b] = dreg;
    }
}

// ---------------------------------------------------------------------------------
// dec_bwd_rows_kernel: the decoder's dX chain (dec_bwd_kernel's three products) on FEW rows, in block_bwd_kernel's form: a 16-wave
// workgroup owns 16 rows, wave w owns hidden tile w of the first two products (latent tile w of the third), the weight fragments
// come straight from the L2-resident images -- for dg2 = s W3^T the K-major image of W3 (rows = hidden tile, k = pixels), streamed
// AD deep over the 25 pixel k-steps -- and s / dpre2 / dpre1 sit in LDS as 1 KiB k-step blocks.  dec_bwd_kernel streams every weight
// unit through ONE workgroup's LDS per 128 rows: at 20 rows that is a single workgroup walking 19 units in turn (27 us).
// ---------------------------------------------------------------------------------
template <int AD>
__global__ __launch_bounds__(1024, (AD > 6 ? 2 : 5)) void dec_bwd_rows_kernel(DecBwdRowsArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * 16;
    const int row = r0 + rho;
    const bool valid = row < a.M;
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    char* act0 = smem;                                   // s tile: KTX blocks
    char* act1 = smem + (size_t)a.KTX * 1024;            // dpre2: KT blocks
    char* act2 = act1 + (size_t)a.KT * 1024;             // dpre1: KT blocks
    float* gx_lds = (float*)(smem + (size_t)(a.KTX + 2 * a.KT) * 1024);      // 16 floats behind the three activation tiles (lse_on)
    if (a.lse_on) {       // wave w: image b0 + w of the (<= 16) images this workgroup's rows belong to (lse_kernel's job, iwae1.py:113-139).
                          // First thing in the kernel: beside the weight fragments' loads it cost 15 spilled registers in the <= 96-register shape.
        const int b0 = r0 / a.lse.k, b1 = min(r0 + 15, a.M - 1) / a.lse.k;
        if (b0 + wave <= b1) {
            LseArgs la = a.lse;
            la.gx_local = gx_lds; la.gx_r0 = r0; la.gx_n = 16;
            lse_image(la, b0 + wave, lane, LseGlobalSrc{});
        }
    }
    for (int c = threadIdx.x; c < a.KTX * 64; c += 1024) {
        const int ks = c >> 6, rr = (c >> 2) & 15, qq = c & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + rr < a.M) v = *(const uint4*)(a.SP + (size_t)(r0 + rr) * a.ldS + ks * 32 + qq * 8);
        *(uint4*)(act0 + ks * 1024 + rr * 64 + ((qq ^ hperm(rr >> 2)) * 16)) = v;
    }
    const int mg = wave >> 2, tg = wave & 3;
    const bool has = wave < a.NT1, has3 = wave < a.NT3;
    const char* w3 = a.imgK3 + (size_t)wave * 1024 + a_off;                    // block (pixel k-step ks, hidden tile wave) = (ks*MT + wave) KiB
    const char* w2 = a.imgB2 + (size_t)mg * img_mg_group_bytes(a.KT) + tg * 1024 + a_off;
    const char* w1 = a.imgB1 + (size_t)mg * img_mg_group_bytes(a.KT) + tg * 1024 + a_off;
    uint4 A2[BLOCKFWD_MAX_KT], A1[BLOCKFWD_MAX_KT];
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks) {
        A2[ks] = make_uint4(0, 0, 0, 0);
        if (ks < a.KT && has) A2[ks] = *(const uint4*)(w2 + (size_t)ks * 4096);
    }
    const int kso = wave >> 1, hh = wave & 1;
    uint2 y2 = make_uint2(0, 0), y1 = make_uint2(0, 0);
    float gxr = 0.0f;
    if (valid && has) {
        y2 = *(const uint2*)(a.G2 + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh);
        y1 = *(const uint2*)(a.G1 + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh);
        if (!a.lse_on) gxr = a.gx[row];
    }
    auto dtanh = [&](const f32x4& acc, const uint2& y, float sc) {
        const float ya = bflo(y.x), yb = bfhi(y.x), yc = bflo(y.y), yd = bfhi(y.y);
        return make_uint2(pack2(sc * acc[0] * (1.0f - ya * ya), sc * acc[1] * (1.0f - yb * yb)), pack2(sc * acc[2] * (1.0f - yc * yc), sc * acc[3] * (1.0f - yd * yd)));
    };
    // ---- product 1: dg2 = s W3^T over the KTX pixel k-steps, A fragments streamed AD deep
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const size_t kstride = (size_t)a.MT3 * 1024;
        uint4 av[AD];
#pragma unroll
        for (int i = 0; i < AD; ++i) av[i] = (i < a.KTX && has) ? *(const uint4*)(w3 + (size_t)i * kstride) : make_uint4(0, 0, 0, 0);
        __syncthreads();                         // s tile staged (and the row weights in LDS)
        for (int k0 = 0; k0 < a.KTX; k0 += AD) {
#pragma unroll
            for (int i = 0; i < AD; ++i) {
                const int ks = k0 + i;
                if (ks < a.KTX) {
                    acc = mfma16(av[i], *(const uint4*)(act0 + ks * 1024 + a_off), acc);
                    if (ks + AD < a.KTX && has) av[i] = *(const uint4*)(w3 + (size_t)(ks + AD) * kstride);
                }
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks) {
        A1[ks] = make_uint4(0, 0, 0, 0);
        if (ks < a.KT && has3) A1[ks] = *(const uint4*)(w1 + (size_t)ks * 4096);
    }
    if (a.lse_on && valid) gxr = gx_lds[rho];      // (written in front of the s-tile barrier above)
    if (has) {
        const uint2 v = valid ? dtanh(acc, y2, gxr) : make_uint2(0, 0);      // dpre2 = g_r * dg2 * (1 - g2^2)
        *(uint2*)(act1 + kso * 1024 + a_off + 8 * hh) = v;
        if (valid) *(uint2*)(a.D2P + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh) = v;
    }
    __syncthreads();
    // ---- product 2: dpre1 = (dpre2 V2^T) * (1 - g1^2)
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
        if (ks < a.KT) acc = mfma16(A2[ks], *(const uint4*)(act1 + ks * 1024 + a_off), acc);
    if (has) {
        const uint2 v = valid ? dtanh(acc, y1, 1.0f) : make_uint2(0, 0);
        *(uint2*)(act2 + kso * 1024 + a_off + 8 * hh) = v;
        if (valid) *(uint2*)(a.D1P + (size_t)row * a.ldH + kso * 32 + q * 8 + 4 * hh) = v;
    }
    __syncthreads();
    // ---- product 3: dz = dpre1 V1^T (latent tile `wave`)
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < BLOCKFWD_MAX_KT; ++ks)
        if (ks < a.KT) acc = mfma16(A1[ks], *(const uint4*)(act2 + ks * 1024 + a_off), acc);
    if (has3 && valid) {
        const int f0 = 16 * wave + 4 * q;
        if (a.DZH) *(uint2*)(a.DZH + (size_t)row * a.ldDZ + f0) = make_uint2(pack2(acc[0], acc[1]), pack2(acc[2], acc[3]));
        else *(float4*)(a.DZ + (size_t)row * a.ldDZ + f0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ---------------------------------------------------------------------------------
// bern_pipe_kernel: the Bernoulli forward (decoder output layer + log p(x|z), iwae1.py:74-75,83,111) at large row counts,
// software-pipelined INSIDE each wave.  dense_kernel<EPI_BERN> runs "MFMAs of a 64-pixel group, then its epilogue": the
// waves of a workgroup pass those phases in lockstep (barriers) and two workgroups sharing a CU fall into step with each
// other (they contend for the same LDS port in one phase and the same VALU in the other), so the MFMA/LDS phase (33 us
// alone) and the epilogue's VALU issue (~80 cycles per logit, 21+ us alone) add up instead of overlapping (52 us).
// Here a wave works in HALF groups (32 pixels = one accumulator tile pair = one bern8 call) and issues the 2*KTC MFMAs of
// half h+1 between the VALU / transcendental instructions of the epilogue of half h (four fenced chunks per half), so
// every wave's own instruction stream keeps both pipes busy and no phase alignment between waves matters.
//   * 8 waves x 16 rows, <= 128 registers (four waves per SIMD): g2 fragments 4*KTC, two accumulator tile pairs 16,
//     two deferred s fragments 8, LDS read pipeline 16
//   * x - 1/2 of the (<= BERN_XIMG_MAX) images the workgroup's 128 rows belong to is staged in LDS once, as fp32: the loop has
//     no global loads, only the weight DMA (one group ahead, issued right behind the barrier) and the deferred s stores
//   * pad halves beyond Np32 are skipped (800 of 832 padded pixels at X = 784)
// ---------------------------------------------------------------------------------
#define BERN_XIMG_MAX 5
// QW: one 16-wave workgroup per CU owning 200 rows = 12.5 sixteen-row tiles (51 200 rows = 256 such workgroups: 3.125 tiles per
// SIMD, no SIMD carries 4).  Waves 0..11 own a full tile each; the half-filled tile 12 belongs to the FOUR waves 12..15
// (one per SIMD): each runs the tanh layers for it (redundantly; wave 12 stores) and takes one of the four 16-pixel tiles of
// EVERY 64-pixel group of the output layer, so a SIMD carries 3.25 tile-equivalents of that phase instead of 4.  Their partial
// log p(x|z) sums meet in LDS.
template <int KTC, bool KEEP, bool PRE, bool QW = false>
__global__ __launch_bounds__(QW ? 1024 : 512, 4) void bern_pipe_kernel(DenseArgs a) {
#ifdef IWAE_DENSE_STAMPS       // diagnostic build: cycles per phase and wave -> a.stamps[wave][8] (prologue + z | layer 1 | layer 2 | fill | main loop | tail | end)
    unsigned long long ds_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ds_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ds_prev)::"memory");
#endif
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NWV = QW ? 16 : 8, ROWS = QW ? 200 : 128;
    constexpr int UNIT = KTC * 4096 + 1024, NPC = 4 * KTC + 1, NIDX = (NPC + NWV - 1) / NWV, NF = 2 * KTC, P = 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int tile = QW ? min(wave, 12) : wave;
    const int qw = QW ? wave - 12 : -1;                 // >= 0: one of the four waves sharing tile 12
    const int row = blockIdx.x * ROWS + tile * 16 + rho;
    const bool valid = row < a.M && tile * 16 + rho < ROWS;
    const bool storer = qw <= 0;                          // quarter waves 1..3 recompute tile 12's activations but do not store them
    // A quarter wave's stage is a short DEPENDENT chain (7 MFMAs into one accumulator, 4 logits per lane, an LDS round trip): sharing
    // its SIMD's issue round-robin with three full waves, it took LONGER per group than they did (phase stamps: 51.6k vs 41.0k cycles
    // of main-loop work) and the full waves waited for it at every group boundary (17k cycles, 14 % of the kernel).  Issued first, it
    // is out of their way instead.
    if (QW && qw >= 0) __builtin_amdgcn_s_setprio(3);
    const int rowc = min(row, a.M - 1);
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    const int H = a.Np32 >> 5;                  // 32-pixel halves that hold real pixels
    const size_t gbytes = img_mg_group_bytes(KTC);

    auto dma_group = [&](int g, int buf) {      // wave w moves pieces w, w+8, ...
#pragma unroll
        for (int idx = 0; idx < NIDX; ++idx) {
            const int p = wave + NWV * idx;
            if (p < NPC)
                glds16(a.img + (size_t)g * gbytes + (size_t)p * 1024 + lane * 16,
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * UNIT) + (uint32_t)p * 1024u)));
        }
    };
    // PRE: the whole decoder in this launch.  The two tanh layers in front of the output layer (iwae1.py:81-82) run first, on
    // the same 16 rows per wave, operands in registers from layer to layer: z (made here from the encoder head and the step's
    // draws -- a.zhead -- or read from pre_Z) -> g1 -> g2, each also stored for the backward pass, g2's fragments ARE the
    // output layer's data operand.  Their weights stream through the same two LDS buffers, one 64-feature group per unit:
    // units 0..MGH-1 layer 1, MGH..2MGH-1 layer 2, then the output layer's group 0 (an even unit: buffer 0, as below).
    constexpr int MGH = (2 * KTC + 3) / 4;
    auto dma_unit = [&](int u, int buf) {
        const char* src = a.img;
        int npc = NPC;
        if (u < MGH) { src = a.pre_img1 + (size_t)u * img_mg_group_bytes(a.pre_KT1); npc = 4 * a.pre_KT1 + 1; }
        else if (u < 2 * MGH) src = a.pre_img2 + (size_t)(u - MGH) * gbytes;
        if (WG_DBG(a, 64) && u < 2 * MGH) return;      // (DIAG ablation, timing only: the tanh layers without their weight stream)
#pragma unroll
        for (int idx = 0; idx < NIDX; ++idx) {
            const int p = wave + NWV * idx;
            if (p < npc)
                glds16(src + (size_t)p * 1024 + lane * 16,
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * UNIT) + (uint32_t)p * 1024u)));
        }
    };
    if constexpr (PRE) dma_unit(0, 0);
    else dma_group(0, 0);
    // x of the images of this block's rows -> LDS (the rows of XB are contiguous)
    // x - 1/2 as fp32 (P order, so a lane's 8 values of a half are 32 contiguous bytes): the epilogue reads its operand ready-made
    char* lx = smem + 2 * UNIT;
    const int blk_r0 = blockIdx.x * ROWS;
    const int b0 = blk_r0 / a.k, b1 = min(blk_r0 + ROWS - 1, a.M - 1) / a.k;
    // Round 5, the 16-wave shape's sampling prologue (z = mu + sigma*eps, iwae1.py:59,107,109).  Counters of round 4's form (per-phase exits,
    // profiles/r05_bern_pipe_phases.txt): 971 vector instructions per wave and 13.8 of the kernel's 63.8 us for 25 z values per lane -- every
    // 4 elements a basic block of its own (masks, a 12-instruction __logf per element) whose three loads were waited for right behind their
    // issue: eight dependent round trips to memory per wave.  Now: the lane's 8 float4 of draws are requested FIRST, all at once (the only
    // HBM stream of the prologue, 20 MB chip-wide); the heads of the workgroup's <= 8 images are staged in LDS once (zero beyond D, so pad
    // features need no masks), sum_d log sigma_d is made ONCE per image by one wave instead of per row and element, and a row's densities are
    //   log p(z) = -1/2 sum z^2 - D/2 log 2pi,   log q(z|x) = -1/2 sum eps^2 - D/2 log 2pi - sum log sigma      ((z - mu)/sigma = eps)
    // -- three fused multiply-adds per element.
    constexpr bool ZQ = PRE && QW;
    constexpr int HZ_OFF = 2 * UNIT + 128 + 4096 + 2 * KTC * 1024 + 4096 + 1024;      // + 8 * ldXB * 4: behind the in-kernel lse_image's area
    float4 ze[8];
    if constexpr (PRE) {
        if (a.zhead) {
            const float* er = a.zeps + (size_t)rowc * a.zldE;
            const int flast = a.zldE - 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) ze[i] = *(const float4*)(er + min(32 * (i >> 1) + 16 * (i & 1) + 4 * q, flast));      // (chunks beyond the row: a valid address, masked below)
        }
    }
    if constexpr (ZQ) {
        if (a.zhead) {
            float* hzf = (float*)(smem + HZ_OFF + (size_t)8 * a.ldXB * 4);
            const int zDp = a.zDp, c4 = zDp >> 2, nimg = b1 - b0 + 1;
            for (int o = threadIdx.x; o < nimg * 2 * c4; o += 64 * NWV) {
                const int img = o / (2 * c4), r = o - img * 2 * c4, arr = r >= c4 ? 1 : 0, c = r - arr * c4;
                const float4 v = *(const float4*)(a.zhead + (size_t)(b0 + img) * a.ldZH + arr * zDp + 4 * c);
                float w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = (4 * c + i < a.zD) ? w[i] : 0.0f;
                *(float4*)(hzf + (size_t)(img * 3 + arr) * zDp + 4 * c) = make_float4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    const int xchunks = (b1 - b0 + 1) * a.ldXB / 8;           // 16-byte pieces of bf16 x (the rows of XB are contiguous)
    for (int o = threadIdx.x; o < xchunks; o += 64 * NWV) {
        const uint4 v = *(const uint4*)(a.XB + (size_t)b0 * a.ldXB + (size_t)o * 8);
        float4 lo, hi;
        lo.x = bflo(v.x) - 0.5f; lo.y = bfhi(v.x) - 0.5f; lo.z = bflo(v.y) - 0.5f; lo.w = bfhi(v.y) - 0.5f;
        hi.x = bflo(v.z) - 0.5f; hi.y = bfhi(v.z) - 0.5f; hi.z = bflo(v.w) - 0.5f; hi.w = bfhi(v.w) - 0.5f;
        *(float4*)(lx + (size_t)o * 32) = lo;
        *(float4*)(lx + (size_t)o * 32 + 16) = hi;
    }
    const char* lxrow = lx + (size_t)(rowc / a.k - b0) * a.ldXB * 4 + q * 32;     // + 128 bytes per half
    uint4 bfr[KTC];
    if constexpr (!PRE) {
#pragma unroll
        for (int ks = 0; ks < KTC; ++ks) {
            const uint4 v = *(const uint4*)(a.X + (size_t)rowc * a.ldX + ks * 32 + q * 8);
            bfr[ks] = valid ? v : make_uint4(0, 0, 0, 0);
        }
    } else {
        const int KT1 = a.pre_KT1, ldZ = 32 * KT1;       // <= 4 k-steps of latent features
        uint4 zf[4];
        if (a.zhead) {
            // z = mu + sigma*eps of this row and its prior / posterior log-densities (iwae1.py:59,107,109).  ONE arithmetic for both workgroup
            // shapes -- a row's densities must not depend on the shape its batch took (two half batches take the 8-wave shape, the full
            // batch the 16-wave one: test_full_size_batch_permutation_and_shard_equivalence) -- they differ in where the heads come from:
            // the 16-wave shape reads its <= 8 images' heads and log-sigma sums from LDS (staged above), the 8-wave shape (80 KiB of LDS
            // for two workgroups per CU: no room) reads the heads from L2 and every wave sums log sigma of its rows' <= 2 images itself.
            const int zDp = a.zDp;
            const int bimg = min(rowc / a.k, b1);      // (the shared 13th tile's rows beyond the workgroup's 200 belong to the next one: any staged image will do, nothing of them is kept)
            const bool dreg = a.zlq_dreg != nullptr;
            auto logsig = [&](const float* sgp, float& s1, float& s2) {      // sum_d log2 sigma_d (and of sigma_d + 1e-6) by one wave: the same reduction wherever it runs
                s1 = 0.0f; s2 = 0.0f;
                for (int d = lane; d < a.zD; d += 64) {
                    const float sg = sgp[d];
                    s1 += log2_raw(sg);
                    if (dreg) s2 += log2_raw(sg + 1e-6f);
                }
                s1 = LN2_F * wave_sum(s1); s2 = LN2_F * wave_sum(s2);
            };
            float hs1 = 0.0f, hs2 = 0.0f;
            const float* hm;                // the row's image: mu at [0, D), sigma at [zDp, zDp + D)
            if constexpr (ZQ) {
                float* hzf = (float*)(smem + HZ_OFF + (size_t)8 * a.ldXB * 4);
                float* hsum = hzf + 8 * 3 * zDp;
                __syncthreads();                  // the images' heads are in LDS
                if (wave <= b1 - b0) {
                    float s1, s2;
                    logsig(hzf + (size_t)(wave * 3 + 1) * zDp, s1, s2);
                    if (lane == 0) { hsum[2 * wave] = s1; hsum[2 * wave + 1] = s2; }
                }
                __syncthreads();
                hm = hzf + (size_t)(bimg - b0) * 3 * zDp;
                hs1 = hsum[2 * (bimg - b0)]; hs2 = hsum[2 * (bimg - b0) + 1];
            } else {
                hm = a.zhead + (size_t)bimg * a.ldZH;
                const int ilo = __builtin_amdgcn_readlane(bimg, 0), ihi = __builtin_amdgcn_readlane(bimg, 15);      // (k >= 32: a tile's 16 rows span <= 2 images)
                float s1, s2;
                logsig(a.zhead + (size_t)ilo * a.ldZH + zDp, s1, s2);
                hs1 = s1; hs2 = s2;
                if (ihi != ilo) {
                    logsig(a.zhead + (size_t)ihi * a.ldZH + zDp, s1, s2);
                    if (bimg != ilo) { hs1 = s1; hs2 = s2; }
                }
            }
            float sz = 0.0f, se = 0.0f, su = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                zf[ks] = make_uint4(0, 0, 0, 0);
                if (ks < KT1) {
                    float z8[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int f0 = 32 * ks + 16 * h + 4 * q;
                        const float4 mu4 = *(const float4*)(hm + f0), sg4 = *(const float4*)(hm + zDp + f0);
                        const float4 e4 = ze[2 * ks + h];
                        float ev[4] = {e4.x, e4.y, e4.z, e4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
                        const float sgv[4] = {sg4.x, sg4.y, sg4.z, sg4.w};
                        if (32 * ks + 16 * h + 16 > a.zD) {      // (wave-uniform: the one or two chunks that straddle or lie beyond D -- pad features are exactly 0)
#pragma unroll
                            for (int i = 0; i < 4; ++i) { const bool in = f0 + i < a.zD; ev[i] = in ? ev[i] : 0.0f; muv[i] = in ? muv[i] : 0.0f; }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float z = fmaf(sgv[i], ev[i], muv[i]);
                            sz = fmaf(z, z, sz);
                            se = fmaf(ev[i], ev[i], se);
                            z8[4 * h + i] = z;
                        }
                        if (dreg) {      // tasks/task02.py:63-65: (z - mu)/(sigma + 1e-6) = eps * sigma/(sigma + 1e-6)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float u2 = (sgv[i] * __builtin_amdgcn_rcpf(sgv[i] + 1e-6f)) * ev[i];
                                su = fmaf(u2, u2, su);
                            }
                        }
                    }
                    const uint4 frag = make_uint4(pack2(z8[0], z8[1]), pack2(z8[2], z8[3]), pack2(z8[4], z8[5]), pack2(z8[6], z8[7]));
                    zf[ks] = valid ? frag : make_uint4(0, 0, 0, 0);
                    if (valid && storer && a.ZPout) *(uint4*)(a.ZPout + (size_t)row * ldZ + ks * 32 + q * 8) = frag;
                }
            }
            sz += __shfl_xor(sz, 16); sz += __shfl_xor(sz, 32);
            se += __shfl_xor(se, 16); se += __shfl_xor(se, 32);
            su += __shfl_xor(su, 16); su += __shfl_xor(su, 32);
            const float cD = 0.5f * LOG2PI_F * (float)a.zD;
            const float lp = -0.5f * sz - cD, lq = -0.5f * se - cD - hs1, lq2 = -0.5f * su - cD - hs2;
            if (q == 0 && valid && storer) { a.zlp[row] = lp; a.zlq[row] = lq; if (dreg) a.zlq_dreg[row] = lq2; }
            if constexpr (QW) {      // (kept in LDS for the in-kernel lse_image at the end: an area nothing else touches)
                if (a.lse_on && q == 0 && storer) {
                    float* lz = (float*)(smem + 2 * UNIT + (size_t)8 * a.ldXB * 4 + 128 + 4096 + 2 * KTC * 1024);
                    lz[tile * 16 + rho] = lp; lz[208 + tile * 16 + rho] = lq; lz[416 + tile * 16 + rho] = lq2;
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                zf[ks] = make_uint4(0, 0, 0, 0);
                if (ks < KT1) {
                    const uint4 v = *(const uint4*)(a.pre_Z + (size_t)rowc * ldZ + ks * 32 + q * 8);
                    zf[ks] = valid ? v : make_uint4(0, 0, 0, 0);
                }
            }
        }
        // one tanh layer: groups of 64 out-features (4 accumulator tiles), data operand bin[0..KTin), result as the next
        // layer's operand fragments bout[] (tile pair 2p, 2p+1 of group mg = k-step 2mg + p) and as P-layout rows in Gout
        uint4 g1f[KTC];
        // Quarter waves (QW) take tile qw of every 64-feature group of BOTH layers for the shared tile 12 -- the units are
        // barrier-separated, so only a per-unit split keeps the four SIMDs level -- and exchange the 8-byte fragment halves
        // through LDS (xch_out; read back as whole fragments behind the next layer's first barrier: xch_in).
        // (round 3: a group's activation stores are issued at the TOP of the next unit, behind its barrier and weight DMA -- issued at the end of their own
        // unit they were what the next unit's wait waited for, eight times per tile; the fragments are the next layer's operand and live anyway.
        // `first`: the stores the unit in front of this layer left behind; this layer's last group is left to the caller)
        auto store_group = [&](uint16_t* Gout, const uint4 (&bout)[KTC], int mg) {
            if (!(valid && storer && Gout) || (QW && qw >= 0)) return;
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
                if (2 * mg + p2 < KTC) *(uint4*)(Gout + (size_t)row * (32 * KTC) + (2 * mg + p2) * 32 + q * 8) = bout[2 * mg + p2];
        };
        // (KTin: an int, or an integral_constant -- the layer's k-step count known at compile time: the run-time form puts a scalar branch
        // around every MFMA and LDS read of the layer, one basic block each)
        auto hidden = [&](auto nb_tag, uint4* bin, auto KTin, int ubase, uint4 (&bout)[KTC], uint16_t* Gout, const char* xch_in, char* xch_out, auto&& first) {
            constexpr int NB = decltype(nb_tag)::value;
            const bool quarter = QW && qw >= 0;
#pragma unroll
            for (int mg = 0; mg < MGH; ++mg) {
                const int u = ubase + mg, buf = u & 1;
                wait_all_vmem();
                __syncthreads();
                dma_unit(u + 1, buf ^ 1);          // behind the last unit of layer 2 comes the output layer's group 0
                if (mg == 0) first(); else store_group(Gout, bout, mg - 1);
                const char* lb = smem + buf * UNIT + a_off;
                const char* lbias = smem + buf * UNIT + KTin * 4096 + q * 16;
                if (quarter) {
                    if (mg == 0 && xch_in) {
#pragma unroll
                        for (int ks = 0; ks < NB; ++ks) bin[ks] = *(const uint4*)(xch_in + ks * 1024 + lane * 16);
                    }
                    const float4 c = *(const float4*)(lbias + 64 * qw);
                    f32x4 acc1 = (f32x4){c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int ks = 0; ks < NB; ++ks)
                        if (ks < KTin) acc1 = mfma16(*(const uint4*)(lb + (ks * 4 + qw) * 1024), bin[ks], acc1);
                    const int ft = 4 * mg + qw, kso = ft >> 1, hh = ft & 1;
                    if (kso < KTC) {
                        const uint2 piece = make_uint2(pack2(tanh_fast(acc1[0]), tanh_fast(acc1[1])), pack2(tanh_fast(acc1[2]), tanh_fast(acc1[3])));
                        *(uint2*)(xch_out + kso * 1024 + lane * 16 + 8 * hh) = valid ? piece : make_uint2(0, 0);
                        if (valid && Gout) *(uint2*)(Gout + (size_t)row * (32 * KTC) + kso * 32 + q * 8 + 4 * hh) = piece;
                    }
                    continue;
                }
                f32x4 acc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 c = *(const float4*)(lbias + 64 * t);
                    acc[t] = (f32x4){c.x, c.y, c.z, c.w};
                }
                lds_pipeline<NB * 4, 4>(
                    [&](int i) { return (i >> 2) < KTin ? *(const uint4*)(lb + i * 1024) : make_uint4(0, 0, 0, 0); },
                    [&](int i, const uint4& av) { if ((i >> 2) < KTin) acc[i & 3] = mfma16(av, bin[i >> 2], acc[i & 3]); });
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int kso = 2 * mg + p2;
                    if (kso < KTC) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = tanh_fast(acc[2 * p2 + (j >> 2)][j & 3]);
                        const uint4 frag = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                        // (rows beyond M / beyond the tile's share carry finite values of a clamped row: a data row is a COLUMN of every product
                        // here, nothing mixes rows, and nothing of such a row is stored -- zeroing them was 8 v_cndmask per group)
                        bout[kso] = frag;      // (stored one unit later: store_group; forward-only calls keep nothing, Gout = null)
                    }
                }
            }
        };
        char* xch1 = smem + 2 * UNIT + (size_t)8 * a.ldXB * 4 + 128 + 4096;      // g1 / g2 of the shared tile: 7 KiB each
        char* xch2 = xch1 + KTC * 1024;
        DS_STAMP(0);
        if (WG_DBG(a, 128)) { wait_all_vmem(); return; }      // (DIAG phase exits, counters only: behind the prologue + z ...
        if (KT1 == 4) hidden(std::integral_constant<int, 4>{}, zf, std::integral_constant<int, 4>{}, 0, g1f, a.pre_G1, nullptr, xch1, [] {});      // (the reference's latent: 100 -> 128 = 4 k-steps)
        else hidden(std::integral_constant<int, 4>{}, zf, KT1, 0, g1f, a.pre_G1, nullptr, xch1, [] {});
        DS_STAMP(1);
        if (WG_DBG(a, 512)) { wait_all_vmem(); return; }      // ... behind the first tanh layer ...
        hidden(std::integral_constant<int, KTC>{}, g1f, std::integral_constant<int, KTC>{}, MGH, bfr, a.pre_G2, xch1, xch2, [&] { store_group(a.pre_G1, g1f, MGH - 1); });
        DS_STAMP(2);
        if (WG_DBG(a, 256)) { wait_all_vmem(); return; }      // ... behind both)
    }

    f32x4 accA[2], accB[2];      // the two tile pairs swap roles every stage (multiply into one, epilogue from the other)
    uint4 st[2];
    float rowacc = 0.0f;
    auto store_s = [&](int h, const uint4& v) {
        if (KEEP && valid && !WG_DBG(a, 32)) *(uint4*)(a.YP + (size_t)row * a.ldYP + 32 * h + 8 * q) = v;
    };
    // One pipeline stage: the 2*KTC MFMAs of half hN = h+1 (tile pair tbN of the group in buffer bufN, accumulators accN,
    // started from the bias block) are issued in four chunks between the epilogue arithmetic of half h (accC, two logits
    // per chunk); sched_barrier fences keep the chunks apart, inside a chunk the compiler schedules freely.  MF = false:
    // epilogue only (last half).  The per-logit arithmetic is bern8's (see there).
    auto stage = [&](auto masked, auto domfma, f32x4 (&accC)[2], f32x4 (&accN)[2], int h, int bufN, int tbN, float4 (&xq)[2], float4 (&xnx)[2], uint4& sp) {
        constexpr bool MASKED = decltype(masked)::value, MF = decltype(domfma)::value;
        const char* lb = smem + bufN * UNIT + a_off + tbN * 1024;
        uint4 av[P];
        if constexpr (MF) {
            const char* lbias = smem + bufN * UNIT + KTC * 4096 + tbN * 64 + q * 16;
            const float4 c0 = *(const float4*)lbias, c1 = *(const float4*)(lbias + 64);
            accN[0] = (f32x4){c0.x, c0.y, c0.z, c0.w};
            accN[1] = (f32x4){c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int i = 0; i < P; ++i) av[i] = *(const uint4*)(lb + ((i >> 1) * 4 + (i & 1)) * 1024);
        }
        float s_xl = 0.0f, s_al = 0.0f, prod = 1.0f, sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int f0 = 32 * h + 4 * q;
        // x - 1/2 of this half was read a stage ago (xq); the next half's is requested here (xnx: the two register sets swap roles
        // from stage to stage, like the accumulators -- a copy back would be 8 v_mov per stage) and pinned at the end of the stage,
        // so no epilogue instruction waits on an LDS read issued right in front of it
        asm volatile("" : "+v"(xq[0].x), "+v"(xq[0].y), "+v"(xq[0].z), "+v"(xq[0].w), "+v"(xq[1].x), "+v"(xq[1].y), "+v"(xq[1].z), "+v"(xq[1].w));
        const float xm8[8] = {xq[0].x, xq[0].y, xq[0].z, xq[0].w, xq[1].x, xq[1].y, xq[1].z, xq[1].w};
        xnx[0] = *(const float4*)(lxrow + (h + 1) * 128); xnx[1] = *(const float4*)(lxrow + (h + 1) * 128 + 16);   // <= one half past the end: the pad
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // chunk fence.  The arithmetic is pure, so a scheduling barrier alone does not hold it in its chunk: the
            // chunk's inputs (two logits, the running accumulators) and the previous chunk's results pass through an
            // empty volatile asm, which pins both ends of every chunk's dependence chains.
            // (the logits pass through it in place, as whole accumulator tiles: pinning copies of the two logits cost 8 v_mov per stage)
            if (MF && c > 0) asm volatile("" : "+v"(accC[0]), "+v"(accC[1]), "+v"(prod), "+v"(s_al), "+v"(s_xl), "+v"(accN[0]), "+v"(accN[1]));
            else asm volatile("" : "+v"(accC[0]), "+v"(accC[1]), "+v"(prod), "+v"(s_al), "+v"(s_xl));
            const float l0 = (c < 2) ? accC[0][(2 * c) & 3] : accC[1][(2 * c) & 3];
            const float l1 = (c < 2) ? accC[0][(2 * c + 1) & 3] : accC[1][(2 * c + 1) & 3];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MF) {     // chunk 0 is arithmetic only: it covers the latency of the LDS reads issued just above
                if (c > 0) {
#pragma unroll
                    for (int i = ((c - 1) * NF) / 3; i < (c * NF) / 3; ++i) {
                        accN[i & 1] = mfma16(av[i % P], bfr[i >> 1], accN[i & 1]);
                        if (i + P < NF) av[i % P] = *(const uint4*)(lb + (((i + P) >> 1) * 4 + ((i + P) & 1)) * 1024);
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * c + jj;
                const float l = jj ? l1 : l0;
                const float xm = xm8[j];
                const float e = exp2_raw(-fabsf(l * LOG2E_F));               // exp(-|l|)
                const bool in = !MASKED || (f0 + 16 * (j >> 2) + (j & 3) < a.Xdim);
                s_al += fabsf(l);
                s_xl = fmaf(xm, l, s_xl);
                if (KEEP) {
                    const float ope = 1.0f + e;
                    prod *= in ? ope : 1.0f;
                    const float hh = __builtin_copysignf(rcp_fast(ope) - 0.5f, l);    // sigmoid(l) - 1/2
                    sv[j] = in ? xm - hh : 0.0f;
                } else {
                    prod = in ? fmaf(prod, e, prod) : prod;                   // prod * (1 + e) in one instruction
                }
            }
            if (KEEP) asm volatile("" : "+v"(sv[2 * c]), "+v"(sv[2 * c + 1]));
        }
        if constexpr (MF) asm volatile("" : "+v"(prod), "+v"(s_al), "+v"(s_xl), "+v"(accN[0]), "+v"(accN[1]));
        asm volatile("" : "+v"(xnx[0].x), "+v"(xnx[0].y), "+v"(xnx[0].z), "+v"(xnx[0].w), "+v"(xnx[1].x), "+v"(xnx[1].y), "+v"(xnx[1].z), "+v"(xnx[1].w));
        __builtin_amdgcn_sched_barrier(0);
        if (KEEP) sp = make_uint4(pack2(sv[0], sv[1]), pack2(sv[2], sv[3]), pack2(sv[4], sv[5]), pack2(sv[6], sv[7]));
        rowacc += s_xl - 0.5f * s_al - LN2_F * log2_raw(prod);
    };

    // MFMAs of one half alone (start of the pipeline)
    auto mfma_only = [&](int buf, int tb, f32x4 (&acc)[2]) {
        const char* lb = smem + buf * UNIT + a_off + tb * 1024;
        const char* lbias = smem + buf * UNIT + KTC * 4096 + tb * 64 + q * 16;
        const float4 c0 = *(const float4*)lbias, c1 = *(const float4*)(lbias + 64);
        acc[0] = (f32x4){c0.x, c0.y, c0.z, c0.w};
        acc[1] = (f32x4){c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int i = 0; i < NF; ++i) acc[i & 1] = mfma16(*(const uint4*)(lb + ((i >> 1) * 4 + (i & 1)) * 1024), bfr[i >> 1], acc[i & 1]);
    };
    // ---- quarter waves (QW): tile qw (16 pixels) of EVERY 64-pixel group, so each SIMD carries the same extra quarter in
    // every group (the groups are barrier-separated: a quarter wave owning whole groups would make its SIMD the slow one
    // of each of them).  One pipeline stage per group, placed behind the group boundary: epilogue of tile qw of group g
    // (4 logits per lane) with the 7 MFMAs of tile qw of group g+1 between its two chunks.
    // (a quarter wave's accumulator tile lives in LDS between groups: as loop-carried registers it cost the full-tile waves
    // of the same kernel spills inside their main loop)
    char* qslot = smem + 2 * UNIT + (size_t)8 * a.ldXB * 4 + 128 + (QW ? max(qw, 0) : 0) * 1024 + lane * 16;
    auto q_real = [&](int g) { return 64 * g + 16 * qw < a.Np32; };
    auto q_mfma = [&](int g) {
        const char* lb = smem + (g & 1) * UNIT + a_off + qw * 1024;
        const float4 c0 = *(const float4*)(smem + (g & 1) * UNIT + KTC * 4096 + qw * 64 + q * 16);
        f32x4 acc = (f32x4){c0.x, c0.y, c0.z, c0.w};
#pragma unroll
        for (int ks = 0; ks < KTC; ++ks) acc = mfma16(*(const uint4*)(lb + ks * 4096), bfr[ks], acc);
        *(f32x4*)qslot = acc;
    };
    auto q_stage = [&](int g, bool mf) {       // epilogue of group g (its logits wait in qslot); mf: then the MFMAs of group g + 1
        const int hq = 2 * g + (qw >> 1), sub = qw & 1;
        const float4 x4 = *(const float4*)(lxrow + hq * 128 + sub * 16);
        const f32x4 lc = *(const f32x4*)qslot;
        const int f0 = 64 * g + 16 * qw + 4 * q;
        const float xm4[4] = {x4.x, x4.y, x4.z, x4.w};
        float s_xl = 0.0f, s_al = 0.0f, prod = 1.0f, sv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float l = lc[j];
            const float e = exp2_raw(-fabsf(l * LOG2E_F));
            const bool in = f0 + j < a.Xdim;
            s_al += fabsf(l);
            s_xl = fmaf(xm4[j], l, s_xl);
            const float ope = 1.0f + e;
            prod *= in ? ope : 1.0f;
            const float hh = __builtin_copysignf(rcp_fast(ope) - 0.5f, l);
            sv[j] = in ? xm4[j] - hh : 0.0f;
        }
        if (KEEP && valid) *(uint2*)(a.YP + (size_t)row * a.ldYP + 32 * hq + 8 * q + 4 * sub) = make_uint2(pack2(sv[0], sv[1]), pack2(sv[2], sv[3]));
        rowacc += s_xl - 0.5f * s_al - LN2_F * log2_raw(prod);
        if (mf) q_mfma(g + 1);
    };
    const bool fullw = !QW || qw < 0;

    wait_all_vmem();
    __syncthreads();
    if (2 < H) dma_group(1, 1);
    if constexpr (PRE) {      // the second tanh layer's last group of activations (deferred by one unit like the others)
        if (valid && storer && a.pre_G2 && !(QW && qw >= 0)) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
                if (2 * (MGH - 1) + p2 < KTC) *(uint4*)(a.pre_G2 + (size_t)row * (32 * KTC) + (2 * (MGH - 1) + p2) * 32 + q * 8) = bfr[2 * (MGH - 1) + p2];
        }
    }
    if constexpr (PRE && QW) {
        if (qw >= 0) {      // the shared tile's g2 fragments, assembled from the four quarter waves' pieces
            const char* xch2 = smem + 2 * UNIT + (size_t)8 * a.ldXB * 4 + 128 + 4096 + KTC * 1024;
#pragma unroll
            for (int ks = 0; ks < KTC; ++ks) bfr[ks] = *(const uint4*)(xch2 + ks * 1024 + lane * 16);
        }
    }
    if (fullw) mfma_only(0, 0, accA);      // half 0: MFMAs only
    else if (q_real(0)) q_mfma(0);
    DS_STAMP(3);

    // at the top of an odd half: half h+1 opens group gN -- its weights must have landed, and group gN-1's buffer is free
    // for group gN+1 once every wave is here; the s fragments of the two previous halves go out now (deferred by a group:
    // a store never sits in front of the next wait)
    auto boundary = [&](int h, bool domf) {
        const int gN = (h + 1) >> 1;
        if (domf) {
            wait_all_vmem();
            __syncthreads();
            if (2 * (gN + 1) < H) dma_group(gN + 1, (gN & 1) ^ 1);
        }
        if (fullw) {
            if (h >= 3) store_s(h - 2, st[1]);
            store_s(h - 1, st[0]);
        }
    };
    float4 xbq[2] = {*(const float4*)lxrow, *(const float4*)(lxrow + 16)}, xbr[2];
    const int Hmain = min(a.Xdim >> 5, H - 1) & ~1;      // halves [0, Hmain): all 32 pixels real, a next half to multiply; in pairs
    int h = 0;
    for (; h < Hmain; h += 2) {
        const int buf = (h >> 1) & 1;
        if (fullw) stage(std::false_type{}, std::true_type{}, accA, accB, h, buf, 2, xbq, xbr, st[0]);          // MFMAs: tiles 2, 3 of this group
        {       // boundary(h + 1, true) with st[1] still holding half h - 1
            DS_STAMP(4);
            wait_all_vmem();
            __syncthreads();
            DS_STAMP(7);
            const int gN = (h >> 1) + 1;
            if (2 * (gN + 1) < H) dma_group(gN + 1, buf);
            if (fullw) {
                if (h >= 2) store_s(h - 1, st[1]);
                store_s(h, st[0]);
            }
        }
        if (fullw) stage(std::false_type{}, std::true_type{}, accB, accA, h + 1, buf ^ 1, 0, xbr, xbq, st[1]);   // tiles 0, 1 of the next group
        else if (q_real(h >> 1)) q_stage(h >> 1, q_real((h >> 1) + 1));
    }
    DS_STAMP(4);
    for (; h < H; ++h) {        // the last halves (masked epilogue; the very last one has nothing left to multiply)
        const int nh = h + 1, bufN = (nh >> 1) & 1, tbN = (nh & 1) * 2;
        const bool domf = nh < H;
        if (h & 1) boundary(h, domf);
        if (!fullw) {
            if ((h & 1) && q_real(h >> 1)) q_stage(h >> 1, domf && q_real((h >> 1) + 1));
            continue;
        }
        // (the tail always multiplies from accA into accB and copies back: the main loop's role swap would cost two more
        // instantiations of the stage here, and their register pressure -- 13 spilled registers in round 1 -- for <= 3 halves)
        uint4 sp = make_uint4(0, 0, 0, 0);
        if (domf) stage(std::true_type{}, std::true_type{}, accA, accB, h, bufN, tbN, xbq, xbr, sp);
        else stage(std::true_type{}, std::false_type{}, accA, accB, h, bufN, tbN, xbq, xbr, sp);
        accA[0] = accB[0]; accA[1] = accB[1];
        xbq[0] = xbr[0]; xbq[1] = xbr[1];
        if (h & 1) st[1] = sp; else st[0] = sp;
    }
    if (fullw) {
        if (H & 1) { if (H >= 2) store_s(H - 2, st[1]); store_s(H - 1, st[0]); }
        else store_s(H - 1, st[1]);
    } else if ((H & 1) && q_real((H - 1) >> 1)) {
        q_stage((H - 1) >> 1, false);       // the last group opened at an even half: its epilogue is still due
    }

    DS_STAMP(5);
    float v = rowacc;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
#ifdef IWAE_DENSE_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        ds_sum[6] = t_ - ds_prev;
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * NWV + wave) * 8 + i] = ds_sum[i];
    }
#endif
    if constexpr (QW) {         // tile 12: the four quarter waves' sums, added in wave order
        __syncthreads();        // (every wave is past its last LDS read: buffer 0 is free)
        float* part = (float*)smem;
        if (qw >= 0 && q == 0) part[qw * 16 + rho] = v;
        __syncthreads();
        if (qw == 0) v = part[rho] + part[16 + rho] + part[32 + rho] + part[48 + rho];
        if (qw > 0 && !(PRE && a.lse_on)) return;
    }
    if (q == 0 && valid && storer) a.lpxz[row] = v;
    if constexpr (QW && PRE) {
        // k divides the workgroup's 200 rows: they are whole images (200 / k of them) -- wave w does for image w what lse_kernel would
        // (same function, same order of operations: bitwise the same results), with the terms this kernel made coming from LDS
        if (a.lse_on) {
            float* lz = (float*)(smem + 2 * UNIT + (size_t)8 * a.ldXB * 4 + 128 + 4096 + 2 * KTC * 1024);
            if (q == 0 && storer) lz[624 + tile * 16 + rho] = v;
            __syncthreads();
            const int nimg = ROWS / a.k, b = blockIdx.x * nimg + wave;
            if (wave < nimg && b < a.B) {
                struct Src {
                    const float* lz; int l0; bool made_z, made_lqd;
                    __device__ __forceinline__ float px(const LseArgs&, int s, int) const { return lz[624 + l0 + s]; }
                    __device__ __forceinline__ float px_total(const LseArgs&, int s, int) const { return lz[624 + l0 + s]; }
                    __device__ __forceinline__ float term(const LseArgs& la, int t, int s, int r) const {
                        return (made_z && t == 1) ? lz[l0 + s] : (made_z && t == 2) ? lz[208 + l0 + s] : la.term[t][r];
                    }
                    __device__ __forceinline__ float lq_dreg(const LseArgs& la, int s, int r) const { return made_lqd ? lz[416 + l0 + s] : la.lq_dreg[r]; }
                };
                LseArgs la = a.lse;
                la.gx_local = lz + 832; la.gx_r0 = blockIdx.x * ROWS; la.gx_n = ROWS;      // (the rows' weights also into LDS: g2w below)
                lse_image(la, b, lane, Src{lz, wave * a.k, a.zhead != nullptr, a.zhead != nullptr && a.zlq_dreg != nullptr});
            }
            if (a.G2W) {
                // round 4: g2w = bf16(g_r * g2) of the tile's rows, from the g2 fragments this wave still holds -- the output layer's weight gradient
                // then runs without row weighting (its loaders' 870 cycles of weighting per stage were what a stage took); pad feature g2w_feat = g_r
                __syncthreads();
                if (valid && storer) {
                    const float gr = lz[832 + tile * 16 + rho];
                    const int pp = p_pos(a.g2w_feat), ksf = pp >> 5, qf = (pp & 31) >> 3, ef = pp & 7;
#pragma unroll
                    for (int ks = 0; ks < KTC; ++ks) {
                        const uint4 f = bfr[ks];
                        uint32_t w[4] = {pack2(bflo(f.x) * gr, bfhi(f.x) * gr), pack2(bflo(f.y) * gr, bfhi(f.y) * gr),
                                         pack2(bflo(f.z) * gr, bfhi(f.z) * gr), pack2(bflo(f.w) * gr, bfhi(f.w) * gr)};
                        if (ks == ksf && q == qf) {
                            const uint32_t gb = pack2(gr, 0.0f) & 0xffffu;
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2)
                                if (e2 == (ef >> 1)) w[e2] = (ef & 1) ? ((w[e2] & 0x0000ffffu) | (gb << 16)) : ((w[e2] & 0xffff0000u) | gb);
                        }
                        *(uint4*)(a.G2W + (size_t)row * (32 * KTC) + ks * 32 + q * 8) = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// chain2_fwd_kernel (round 3): the per-sample blocks of the 2-layer model's forward pass in ONE launch at large row counts --
//   z1 = mu1 + sigma1*eps1 with log q(z1|x) (iwae2.py:61, :123),
//   q(z2|z1) = BasicBlock(z1) (iwae2.py:63-64), z2 = mu2 + sigma2*eps2 with log q(z2|z1) and log p(z2) (:65, :119, :124),
//   p(z1|z2) = BasicBlock(z2) (iwae2.py:90) and log p(z1|z2) (:122).
// As dense_kernel launches these were six GEMM launches + sample_kernel + gauss_lp_kernel with every activation and both float32
// heads (1.5 KB per row) making a round trip through HBM between them (~130 us of kernel time at 51 200 rows).  Here a wave owns
// 16 rows through all six layers: a layer's converted accumulators ARE the next layer's B operand (layout.h), the mu accumulators
// wait in registers for their sigma group, z2 and the three log-densities are made in the epilogue that already holds their
// operands.  The weights stream through two LDS buffers one 64-out-feature group at a time (units 0..: l1, l2, head of the encode
// block, then of the decode block), each unit's DMA issued behind the previous unit's barrier.  The hidden activations (and, for
// the unfused backward kernels, the heads) are stored once for the backward pass; a forward-only call stores nothing but z2.
// Shapes: latent widths padded to 64-feature groups (KT0, KT1 even), one hidden width KTH for both blocks.
// ---------------------------------------------------------------------------------
template <int KT0, int KTH, int KT1>
__global__ __launch_bounds__(512, 4) void chain2_fwd_kernel(Chain2FwdArgs a) {
    static_assert(KT0 % 2 == 0 && KT1 % 2 == 0 && KT0 <= 4 && KTH <= 4 && KT1 <= 4, "64-feature latent groups, <= 128 features everywhere");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NWV = 8;
    constexpr int KTMAX = KT0 > KTH ? (KT0 > KT1 ? KT0 : KT1) : (KTH > KT1 ? KTH : KT1);
    constexpr int UNIT = KTMAX * 4096 + 1024, NIDX = (4 * KTMAX + 1 + NWV - 1) / NWV;
    constexpr int MGH = (KTH + 1) / 2;                       // 64-feature groups of a hidden layer
    constexpr int U_E2 = MGH, U_EH = 2 * MGH, U_D1 = U_EH + KT1, U_D2 = U_D1 + MGH, U_DH = U_D2 + MGH, NUNITS = U_DH + KT0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * NWV + wave) * 16 + rho;
    const bool valid = row < a.M;
    const int rowc = min(row, a.M - 1);
    const int b = rowc / a.k, sidx = rowc - b * a.k;
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);

    auto dma_unit = [&](int uu, int buf) {                 // uu wave-uniform
        const char* src; int kt;
        if (uu < U_E2) { src = a.e_img1 + (size_t)uu * img_mg_group_bytes(KT0); kt = KT0; }
        else if (uu < U_EH) { src = a.e_img2 + (size_t)(uu - U_E2) * img_mg_group_bytes(KTH); kt = KTH; }
        else if (uu < U_D1) { src = a.e_imgh + (size_t)(uu - U_EH) * img_mg_group_bytes(KTH); kt = KTH; }
        else if (uu < U_D2) { src = a.d_img1 + (size_t)(uu - U_D1) * img_mg_group_bytes(KT1); kt = KT1; }
        else if (uu < U_DH) { src = a.d_img2 + (size_t)(uu - U_D2) * img_mg_group_bytes(KTH); kt = KTH; }
        else { src = a.d_imgh + (size_t)(uu - U_DH) * img_mg_group_bytes(KTH); kt = KTH; }
        const int npc = 4 * kt + 1;
#pragma unroll
        for (int idx = 0; idx < NIDX; ++idx) {
            const int p = wave + NWV * idx;
            if (p < npc)
                glds16(src + (size_t)p * 1024 + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * UNIT) + (uint32_t)p * 1024u)));
        }
    };
    dma_unit(0, 0);
    auto quad_sum = [&](float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };
    // ---- z1 = mu1 + sigma1 * eps1 of this row (iwae2.py:61) and log q(z1|x) (:123): sample_kernel's arithmetic, without its launch and
    // without reading z1 back -- the fragments are this kernel's first operand; the rows are kept for the decoder and the weight gradient
    // (round 5: the three sampling / density sections of this kernel in the decoder kernel's new form -- every load of a section requested before
    // the first use instead of a load-and-wait per 4 elements, no per-element branches (the chunks that straddle or lie beyond the latent width
    // take a wave-uniform masked path), (z - mu)/sigma = eps for a density scored at its own sample, and sum log sigma as log2 of pair
    // products: one v_log per 2 elements instead of a 12-instruction __logf per element.  Counters before: 3 578 vector instructions per wave
    // for 208 MFMAs, profiles/r05_c2_*.)
    auto log2_pairs = [&](const float (&sv)[4]) { return log2_raw(sv[0] * sv[1]) + log2_raw(sv[2] * sv[3]); };
    uint4 zf[KT0];
    {
        const float* hz = a.head1 + (size_t)b * a.ldH1;
        float e[2 * KT0][4];
        float4 mu4[2 * KT0], sg4[2 * KT0];
        auto request = [&](int c) {
            const int f0 = 32 * (c >> 1) + 16 * (c & 1) + 4 * q;
            e[c][0] = e[c][1] = e[c][2] = e[c][3] = 0.0f;
            if (f0 < a.D0) eps4(a.eps1, b, sidx, rowc, f0 >> 2, a.D0, e[c]);
            mu4[c] = *(const float4*)(hz + f0); sg4[c] = *(const float4*)(hz + 32 * KT0 + f0);      // (f0 < 32 KT0: inside the padded head row)
        };
        // (two k-steps = four chunks = twelve 16-byte loads at a time: all eight chunks at once are 96 registers of loads in flight and spilled)
#pragma unroll
        for (int c = 0; c < 4 && c < 2 * KT0; ++c) request(c);
        float se = 0.0f, sl = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KT0; ++ks) {
            if (ks == 2) {
#pragma unroll
                for (int c = 4; c < 2 * KT0; ++c) request(c);
            }
            float z8[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 2 * ks + h, f0 = 32 * ks + 16 * h + 4 * q;
                float muv[4] = {mu4[c].x, mu4[c].y, mu4[c].z, mu4[c].w}, sgv[4] = {sg4[c].x, sg4[c].y, sg4[c].z, sg4[c].w};
                if (32 * ks + 16 * h + 16 > a.D0) {      // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const bool in = f0 + i < a.D0; e[c][i] = in ? e[c][i] : 0.0f; muv[i] = in ? muv[i] : 0.0f; sgv[i] = in ? sgv[i] : 1.0f; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    z8[4 * h + i] = fmaf(sgv[i], e[c][i], muv[i]);              // iwae2.py:61
                    se = fmaf(e[c][i], e[c][i], se);
                }
                sl += log2_pairs(sgv);
            }
            const uint4 frag = make_uint4(pack2(z8[0], z8[1]), pack2(z8[2], z8[3]), pack2(z8[4], z8[5]), pack2(z8[6], z8[7]));
            zf[ks] = frag;      // (a data row is a COLUMN of every product here: rows beyond M carry a clamped row's finite values, nothing of them is stored)
            if (valid) *(uint4*)(a.Z1P + (size_t)row * (32 * KT0) + ks * 32 + q * 8) = frag;
        }
        se = quad_sum(se); sl = quad_sum(sl);
        if (q == 0 && valid) a.lqz1x[row] = -0.5f * se - 0.5f * LOG2PI_F * (float)a.D0 - LN2_F * sl;      // iwae2.py:123
    }
    int u = 0;
    // one unit: the 4 accumulator tiles (64 out-features x 16 rows) of a weight group, started from the group's bias block
    auto unit_mfma = [&](auto kt_tag, const uint4* bin, f32x4 (&acc)[4]) {
        constexpr int KTin = decltype(kt_tag)::value;
        const int buf = u & 1;
        wait_all_vmem();
        __syncthreads();
        if (u + 1 < NUNITS) dma_unit(u + 1, buf ^ 1);
        const char* lb = smem + buf * UNIT + a_off;
        const char* lbias = smem + buf * UNIT + KTin * 4096 + q * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 c = *(const float4*)(lbias + 64 * t);
            acc[t] = (f32x4){c.x, c.y, c.z, c.w};
        }
        lds_pipeline<KTin * 4, 4>([&](int i) { return *(const uint4*)(lb + i * 1024); },
                                  [&](int i, const uint4& av) { acc[i & 3] = mfma16(av, bin[i >> 2], acc[i & 3]); });
        ++u;
    };
    // a tanh layer: MGH units; tile pair (2p, 2p+1) of group mg is k-step 2mg + p of the next layer's operand (and of the stored rows)
    auto tanh_layer = [&](auto kt_tag, const uint4* bin, uint4 (&bout)[KTH], uint16_t* Hout) {
#pragma unroll
        for (int mg = 0; mg < MGH; ++mg) {
            f32x4 acc[4];
            unit_mfma(kt_tag, bin, acc);
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int kso = 2 * mg + p2;
                if (kso < KTH) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = tanh_fast(acc[2 * p2 + (j >> 2)][j & 3]);
                    const uint4 frag = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                    bout[kso] = frag;
                    if (valid && Hout) *(uint4*)(Hout + (size_t)row * (32 * KTH) + kso * 32 + q * 8) = frag;
                }
            }
        }
    };

    // ---- q(z2|z1): encode_z1_to_z2 (iwae2.py:63-64)
    uint4 h1f[KTH], h2f[KTH];
    tanh_layer(std::integral_constant<int, KT0>{}, zf, h1f, a.EH1);
    tanh_layer(std::integral_constant<int, KTH>{}, h1f, h2f, a.EH2);
    // head: the mu groups first (their accumulators wait), then the sigma groups; z2 = mu2 + sigma2 * eps2 (iwae2.py:65)
    uint4 z2f[KT1];
    {
        constexpr int NG = KT1 / 2;               // 64-feature groups of mu2 (and of sigma2)
        f32x4 mu2[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) unit_mfma(std::integral_constant<int, KTH>{}, h2f, mu2[g]);
        float sz = 0.0f, se = 0.0f, sl = 0.0f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float e[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {      // the group's draws, requested in front of its sigma unit
                const int f0 = 64 * g + 16 * t + 4 * q;
                e[t][0] = e[t][1] = e[t][2] = e[t][3] = 0.0f;
                if (f0 < a.D1) eps4(a.eps2, b, sidx, rowc, f0 >> 2, a.D1, e[t]);
            }
            f32x4 sa[4];
            unit_mfma(std::integral_constant<int, KTH>{}, h2f, sa);
            float zt[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f0 = 64 * g + 16 * t + 4 * q;
                float sgv[4], muv[4] = {mu2[g][t][0], mu2[g][t][1], mu2[g][t][2], mu2[g][t][3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) sgv[i] = exp2_raw(sa[t][i] * LOG2E_F) + 1e-6f;          // iwae2.py:43 (exp activation), :45 (+ 1e-6)
                if (valid && a.EHEAD) {
                    *(float4*)(a.EHEAD + (size_t)row * (64 * KT1) + f0) = make_float4(muv[0], muv[1], muv[2], muv[3]);
                    *(float4*)(a.EHEAD + (size_t)row * (64 * KT1) + 32 * KT1 + f0) = make_float4(sgv[0], sgv[1], sgv[2], sgv[3]);
                }
                if (64 * g + 16 * t + 16 > a.D1) {      // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const bool in = f0 + i < a.D1; e[t][i] = in ? e[t][i] : 0.0f; muv[i] = in ? muv[i] : 0.0f; sgv[i] = in ? sgv[i] : 1.0f; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = fmaf(sgv[i], e[t][i], muv[i]);      // iwae2.py:65
                    sz = fmaf(z, z, sz);                                // log p(z2), iwae2.py:119
                    se = fmaf(e[t][i], e[t][i], se);                    // log q(z2|z1), iwae2.py:124: (z2 - mu2)/sigma2 = eps2
                    zt[t][i] = z;
                }
                sl += log2_pairs(sgv);
            }
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const uint4 frag = make_uint4(pack2(zt[2 * p2][0], zt[2 * p2][1]), pack2(zt[2 * p2][2], zt[2 * p2][3]),
                                              pack2(zt[2 * p2 + 1][0], zt[2 * p2 + 1][1]), pack2(zt[2 * p2 + 1][2], zt[2 * p2 + 1][3]));
                z2f[2 * g + p2] = frag;
                if (valid && a.Z2P) *(uint4*)(a.Z2P + (size_t)row * (32 * KT1) + (2 * g + p2) * 32 + q * 8) = frag;
            }
        }
        sz = quad_sum(sz); se = quad_sum(se); sl = quad_sum(sl);
        const float cD = 0.5f * LOG2PI_F * (float)a.D1;
        if (q == 0 && valid) { a.lpz2[row] = -0.5f * sz - cD; a.lqz2z1[row] = -0.5f * se - cD - LN2_F * sl; }
    }
    // ---- p(z1|z2): decode_z2_to_z1 (iwae2.py:90) and log p(z1|z2) (iwae2.py:122), z1 = mu1 + sigma1 * eps1 in float32
    uint4 g1f[KTH], g2f[KTH];
    tanh_layer(std::integral_constant<int, KT1>{}, z2f, g1f, a.DH1);
    tanh_layer(std::integral_constant<int, KTH>{}, g1f, g2f, a.DH2);
    {
        constexpr int NG = KT0 / 2;
        f32x4 mup[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) unit_mfma(std::integral_constant<int, KTH>{}, g2f, mup[g]);
        float su = 0.0f, sl = 0.0f;
        const float* hz = a.head1 + (size_t)b * a.ldH1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // z1 = mu1 + sigma1 * eps1 once more, in float32 (its bf16 fragments went into the first layer): the group's draws and the image's head,
            // requested in front of the sigma unit
            float e[4][4];
            float4 zm[4], zs[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f0 = 64 * g + 16 * t + 4 * q;
                e[t][0] = e[t][1] = e[t][2] = e[t][3] = 0.0f;
                if (f0 < a.D0) eps4(a.eps1, b, sidx, rowc, f0 >> 2, a.D0, e[t]);
            }
            f32x4 sa[4];
            unit_mfma(std::integral_constant<int, KTH>{}, g2f, sa);
            // (the head behind the unit, two tiles at a time: with the draws it would be 48 registers held across the MFMAs, beside the mu accumulators)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f0 = 64 * g + 16 * t + 4 * q;
                if ((t & 1) == 0) {
#pragma unroll
                    for (int tt = t; tt < t + 2; ++tt) {
                        const int ff = 64 * g + 16 * tt + 4 * q;
                        zm[tt] = *(const float4*)(hz + ff); zs[tt] = *(const float4*)(hz + 32 * KT0 + ff);
                    }
                }
                const float zmv[4] = {zm[t].x, zm[t].y, zm[t].z, zm[t].w}, zsv[4] = {zs[t].x, zs[t].y, zs[t].z, zs[t].w};
                float sgv[4], uv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sgv[i] = exp2_raw(sa[t][i] * LOG2E_F) + 1e-6f;
                    const float z = fmaf(zsv[i], e[t][i], zmv[i]);
                    uv[i] = (z - mup[g][t][i]) * __builtin_amdgcn_rcpf(sgv[i]);      // iwae2.py:122
                }
                if (valid && a.DHEAD) {
                    *(float4*)(a.DHEAD + (size_t)row * (64 * KT0) + f0) = make_float4(mup[g][t][0], mup[g][t][1], mup[g][t][2], mup[g][t][3]);
                    *(float4*)(a.DHEAD + (size_t)row * (64 * KT0) + 32 * KT0 + f0) = make_float4(sgv[0], sgv[1], sgv[2], sgv[3]);
                }
                if (64 * g + 16 * t + 16 > a.D0) {      // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const bool in = f0 + i < a.D0; uv[i] = in ? uv[i] : 0.0f; sgv[i] = in ? sgv[i] : 1.0f; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) su = fmaf(uv[i], uv[i], su);
                sl += log2_pairs(sgv);
            }
        }
        su = quad_sum(su); sl = quad_sum(sl);
        if (q == 0 && valid) a.lpz1z2[row] = -0.5f * su - 0.5f * LOG2PI_F * (float)a.D0 - LN2_F * sl;
    }
}

// ---------------------------------------------------------------------------------
// gblock_bwd_kernel (round 3): the backward pass of one per-sample BasicBlock of the 2-layer model, FROM its Gaussian head, in one launch
// at large row counts.  As separate launches this was gauss_bwd_kernel (reading the stored float32 head, 0.5-1 KB per row) + three
// dense_kernel<EPI_DX / EPI_F32> launches with dhead / dpre2 / dpre1 / dz making round trips through HBM.  Here a wave owns 16 rows:
//   A. the head is RECOMPUTED from the stored h2 (KTL weight groups through the matrix pipe: cheaper than storing and re-reading it --
//      the forward pass no longer writes the heads at all), and its gradient made in that epilogue:
//        MODE 0, decode_z2_to_z1 (iwae2.py:90,122): u = (z1 - mu_p)/sigma_p,  dhead = G [ u/sigma_p | (u^2 - 1)/sigma_p * (sigma_p - 1e-6) ],
//                the direct term of dz1, -G u/sigma_p, kept as bf16 for MODE 1's launch;
//        MODE 1, encode_z1_to_z2 (iwae2.py:63-65,119,124): dz2 = dz2_dec - G z2,  dhead = [ dz2 | (dz2 eps2 + G/sigma2)(sigma2 - 1e-6) ];
//   B-D. the dX chain d2 = (dhead Wh^T)(1 - h2^2), d1 = (d2 W2^T)(1 - h1^2), dz_in = d1 W1^T with every converted accumulator the next
//      product's B operand (layout.h); dhead, d2, d1 are stored once (bf16) for the weight gradients.  MODE 1 adds the other two terms of
//      dz1 (the decoder's, float32, and MODE 0's) to its own and stores the sum as bf16 -- latent_bwd_kernel then reads one tensor, not three.
// Weights: the forward head image, then the three backward images, one 64-feature group per unit through two LDS buffers.
// ---------------------------------------------------------------------------------
template <int MODE, int KTIN, int KTH, int KTL>
__global__ __launch_bounds__(512, 4) void gblock_bwd_kernel(GBlockBwdArgs a) {
    static_assert(KTIN % 2 == 0 && KTL % 2 == 0 && KTIN <= 4 && KTH <= 4 && KTL <= 4, "64-feature latent groups, <= 128 features");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NWV = 8, KTD = 2 * KTL;                    // k-steps of dhead
    constexpr int KTMAX = KTD > KTH ? KTD : KTH;
    constexpr int UNIT = KTMAX * 4096 + 1024, NIDX = (4 * KTMAX + 1 + NWV - 1) / NWV;
    constexpr int MGH = (KTH + 1) / 2, NGL = KTL / 2, NGI = KTIN / 2;
    constexpr int U_B = KTL, U_C = U_B + MGH, U_D = U_C + MGH, NUNITS = U_D + NGI;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * NWV + wave) * 16 + rho;
    const bool valid = row < a.M;
    const int rowc = min(row, a.M - 1);
    const int b = rowc / a.k, sidx = rowc - b * a.k;
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);

    auto dma_unit = [&](int uu, int buf) {                 // uu wave-uniform
        const char* src; int kt;
        if (uu < U_B) { src = a.imgH + (size_t)uu * img_mg_group_bytes(KTH); kt = KTH; }
        else if (uu < U_C) { src = a.imgBh + (size_t)(uu - U_B) * img_mg_group_bytes(KTD); kt = KTD; }
        else if (uu < U_D) { src = a.imgB2 + (size_t)(uu - U_C) * img_mg_group_bytes(KTH); kt = KTH; }
        else { src = a.imgB1 + (size_t)(uu - U_D) * img_mg_group_bytes(KTH); kt = KTH; }
        const int npc = 4 * kt + 1;
#pragma unroll
        for (int idx = 0; idx < NIDX; ++idx) {
            const int p = wave + NWV * idx;
            if (p < npc)
                glds16(src + (size_t)p * 1024 + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * UNIT) + (uint32_t)p * 1024u)));
        }
    };
    dma_unit(0, 0);
    const float G = valid ? a.gx[rowc] : 0.0f;
    uint4 hf2[KTH];
#pragma unroll
    for (int ks = 0; ks < KTH; ++ks) {
        const uint4 v = *(const uint4*)(a.H2 + (size_t)rowc * (32 * KTH) + ks * 32 + q * 8);
        hf2[ks] = valid ? v : make_uint4(0, 0, 0, 0);
    }
    int u = 0;
    // One unit: wait + barrier | weight DMA of the next unit | `after` (the stores the PREVIOUS unit's epilogue left behind and the loads
    // THIS unit's epilogue will want: a store or a load issued here has a whole unit before the next wait) | MFMAs.
    auto unit_mfma = [&](auto kt_tag, auto bias_tag, const uint4* bin, f32x4 (&acc)[4], auto&& after) {
        constexpr int KTin = decltype(kt_tag)::value;
        const int buf = u & 1;
        wait_all_vmem();
        __syncthreads();
        if (u + 1 < NUNITS) dma_unit(u + 1, buf ^ 1);
        after();
        const char* lb = smem + buf * UNIT + a_off;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (decltype(bias_tag)::value) {
                const float4 c = *(const float4*)(smem + buf * UNIT + KTin * 4096 + q * 16 + 64 * t);
                acc[t] = (f32x4){c.x, c.y, c.z, c.w};
            } else acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
        lds_pipeline<KTin * 4, 4>([&](int i) { return *(const uint4*)(lb + i * 1024); },
                                  [&](int i, const uint4& av) { acc[i & 3] = mfma16(av, bin[i >> 2], acc[i & 3]); });
        ++u;
    };
    // ---- A. the head again (mu groups, then sigma groups) and its gradient
    uint4 dhf[KTD];
    {
        f32x4 muh[NGL][4];
#pragma unroll
        for (int g = 0; g < NGL; ++g) unit_mfma(std::integral_constant<int, KTH>{}, std::true_type{}, hf2, muh[g], [] {});
        const float* hz = MODE == 0 ? a.head1 + (size_t)b * a.ldH1 : nullptr;
#pragma unroll
        for (int g = 0; g < NGL; ++g) {
            f32x4 sa[4];
            // the epilogue's operands (draws, the image's head or the decode block's dz2) are requested in front of the unit's MFMAs,
            // and the fragments the previous group's epilogue made are stored there (not behind it, in front of the next wait)
            float ev[4][4];
            float4 xq[4], yq[4];
            unit_mfma(std::integral_constant<int, KTH>{}, std::true_type{}, hf2, sa, [&] {
                if (g > 0 && valid) {
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        *(uint4*)(a.DHP + (size_t)row * (32 * KTD) + (2 * (g - 1) + p2) * 32 + q * 8) = dhf[2 * (g - 1) + p2];
                        *(uint4*)(a.DHP + (size_t)row * (32 * KTD) + (KTL + 2 * (g - 1) + p2) * 32 + q * 8) = dhf[KTL + 2 * (g - 1) + p2];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int f0 = 64 * g + 16 * t + 4 * q;
                    ev[t][0] = ev[t][1] = ev[t][2] = ev[t][3] = 0.0f;
                    xq[t] = make_float4(0.f, 0.f, 0.f, 0.f); yq[t] = xq[t];
                    if (f0 < a.D) {
                        eps4(a.eps, b, sidx, rowc, f0 >> 2, a.D, ev[t]);
                        if (MODE == 1) xq[t] = *(const float4*)(a.DZIN + (size_t)rowc * (32 * KTL) + f0);      // dz2 from the decode block
                    }
                }
            });
            float dmv[4][4], dsv[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f0 = 64 * g + 16 * t + 4 * q;
                const float* e = ev[t];
                float4 x4 = xq[t], y4 = yq[t];
                // (MODE 0: mu1, sigma1 of the row's image -- a few KB per workgroup, cache-resident -- are read here: 32 more registers
                // held across the MFMAs spilled)
                if (MODE == 0 && f0 < a.D) { x4 = *(const float4*)(hz + f0); y4 = *(const float4*)(hz + 32 * KTL + f0); }
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
                float dzd[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float dm = 0.0f, ds = 0.0f;
                    if (f0 + i < a.D) {
                        const float mu = muh[g][t][i], sg = exp2_raw(sa[t][i] * LOG2E_F) + 1e-6f, rs = __builtin_amdgcn_rcpf(sg);
                        if (MODE == 0) {
                            const float z = xv[i] + yv[i] * e[i];
                            const float uu = (z - mu) * rs;
                            dm = G * uu * rs;
                            ds = G * (uu * uu - 1.0f) * rs * (sg - 1e-6f);
                            dzd[i] = -dm;
                        } else {
                            const float z = mu + sg * e[i];
                            const float d = xv[i] - G * z;
                            dm = d;
                            ds = (d * e[i] + G * rs) * (sg - 1e-6f);
                        }
                    }
                    dmv[t][i] = dm; dsv[t][i] = ds;
                }
                if (MODE == 0 && valid) *(uint2*)(a.DZD + (size_t)row * (32 * KTL) + f0) = make_uint2(pack2(dzd[0], dzd[1]), pack2(dzd[2], dzd[3]));
            }
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const uint4 fm = make_uint4(pack2(dmv[2 * p2][0], dmv[2 * p2][1]), pack2(dmv[2 * p2][2], dmv[2 * p2][3]),
                                            pack2(dmv[2 * p2 + 1][0], dmv[2 * p2 + 1][1]), pack2(dmv[2 * p2 + 1][2], dmv[2 * p2 + 1][3]));
                const uint4 fs = make_uint4(pack2(dsv[2 * p2][0], dsv[2 * p2][1]), pack2(dsv[2 * p2][2], dsv[2 * p2][3]),
                                            pack2(dsv[2 * p2 + 1][0], dsv[2 * p2 + 1][1]), pack2(dsv[2 * p2 + 1][2], dsv[2 * p2 + 1][3]));
                dhf[2 * g + p2] = valid ? fm : make_uint4(0, 0, 0, 0);
                dhf[KTL + 2 * g + p2] = valid ? fs : make_uint4(0, 0, 0, 0);
            }
        }
    }
    // a dX product with the tanh' of the stored activation: dout = (din W^T) * (1 - act^2)
    // (stores deferred by one unit: `first` = what the unit in front of this layer left behind; this layer's last group is left to the caller)
    auto store_frags = [&](uint16_t* Dst, const uint4* f, int mg) {
        if (!valid) return;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
            if (2 * mg + p2 < KTH) *(uint4*)(Dst + (size_t)row * (32 * KTH) + (2 * mg + p2) * 32 + q * 8) = f[2 * mg + p2];
    };
    auto dx_layer = [&](auto kt_tag, const uint4* din, const uint4* act, uint4 (&dout)[KTH], uint16_t* Dst, auto&& first) {
#pragma unroll
        for (int mg = 0; mg < MGH; ++mg) {
            f32x4 acc[4];
            unit_mfma(kt_tag, std::false_type{}, din, acc, [&] { if (mg == 0) first(); else store_frags(Dst, dout, mg - 1); });
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int kso = 2 * mg + p2;
                if (kso < KTH) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float y = bf_at(act[kso], j); v[j] = acc[2 * p2 + (j >> 2)][j & 3] * (1.0f - y * y); }
                    const uint4 frag = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                    dout[kso] = valid ? frag : make_uint4(0, 0, 0, 0);
                }
            }
        }
    };
    // ---- B. d2 = (dhead Wh^T)(1 - h2^2)
    uint4 d2f[KTH], hf1[KTH], d1f[KTH];
    dx_layer(std::integral_constant<int, KTD>{}, dhf, hf2, d2f, a.D2P, [&] {
        if (valid) {       // the last head group's fragments
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                *(uint4*)(a.DHP + (size_t)row * (32 * KTD) + (2 * (NGL - 1) + p2) * 32 + q * 8) = dhf[2 * (NGL - 1) + p2];
                *(uint4*)(a.DHP + (size_t)row * (32 * KTD) + (KTL + 2 * (NGL - 1) + p2) * 32 + q * 8) = dhf[KTL + 2 * (NGL - 1) + p2];
            }
        }
#pragma unroll
        for (int ks = 0; ks < KTH; ++ks) {      // h1 for the layer after this one: requested two units ahead
            const uint4 v = *(const uint4*)(a.H1 + (size_t)rowc * (32 * KTH) + ks * 32 + q * 8);
            hf1[ks] = valid ? v : make_uint4(0, 0, 0, 0);
        }
    });
    // ---- C. d1 = (d2 W2^T)(1 - h1^2)
    dx_layer(std::integral_constant<int, KTH>{}, d2f, hf1, d1f, a.D1P, [&] { store_frags(a.D2P, d2f, MGH - 1); });
    // ---- D. gradient of the block's input: dz = d1 W1^T
#pragma unroll
    for (int g = 0; g < NGI; ++g) {
        f32x4 acc[4];
        float4 ddq[4];
        uint2 drq[4];
        unit_mfma(std::integral_constant<int, KTH>{}, std::false_type{}, d1f, acc, [&] {
            if (g == 0) store_frags(a.D1P, d1f, MGH - 1);
            if (MODE == 1) {      // the other two terms of dz1, requested in front of the MFMAs
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int f0 = 64 * g + 16 * t + 4 * q;
                    ddq[t] = *(const float4*)(a.DZDEC + (size_t)rowc * a.ldDZDEC + f0);
                    drq[t] = *(const uint2*)(a.DZD + (size_t)rowc * (32 * KTIN) + f0);
                }
            }
        });
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int f0 = 64 * g + 16 * t + 4 * q;
            if (!valid) continue;
            if (MODE == 0) {
                *(float4*)(a.DZ2 + (size_t)row * (32 * KTIN) + f0) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            } else {       // dz1 = decoder term + direct p(z1|z2) term + this path through q(z2|z1)
                const float4 dd = ddq[t];
                const uint2 dr = drq[t];
                *(uint2*)(a.DZOUT + (size_t)row * (32 * KTIN) + f0) =
                    make_uint2(pack2(acc[t][0] + dd.x + bflo(dr.x), acc[t][1] + dd.y + bfhi(dr.x)), pack2(acc[t][2] + dd.z + bflo(dr.y), acc[t][3] + dd.w + bfhi(dr.y)));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// out_bwd_kernel: backward of the Bernoulli output layer for one block of rows, logits
// recomputed on the fly (never stored): per 64-pixel group
//   l = g2 V3 + c3 ; dl = gx[row] * (x - sigmoid(l))  -> bf16 (also stored, P-layout, for dV3)
//   dg2 += dl V3^T   (the dl accumulator IS the B operand, no LDS round trip)
// then dpre2 = dg2 * (1 - g2^2), P-layout.   (iwae1.py:74-75,111,159)
// One wave per SIMD (4 waves, 32 rows each): 16x2 f32x4 accumulators for dg2 stay resident.
// Same software pipeline as dense_kernel (x prefetched one group ahead, dl stores deferred).
// ---------------------------------------------------------------------------------
// STAMPS = true is a diagnostic build only (s_memtime per phase, summed per wave into a.stamps;
// nothing is computed from the stamps).  The product path launches STAMPS = false.
#define OB_STAMP(slot)                                                                         \
    if (STAMPS) {                                                                              \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        tsum[slot] += t_ - tprev;                                                              \
        tprev = t_;                                                                            \
    }
template <int KTC, bool STAMPS>
__global__ __launch_bounds__(256, 1) void out_bwd_kernel(OutBwdArgs a) {
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    const int KT = KTC ? KTC : a.KT, MT2 = 2 * KT;
    const int u1 = KT * 4096 + 1024;           // W^T group incl. its bias block
    const int unit = u1 + 2 * MT2 * 1024;      // + W k-group
    int row[2], bidx[2];
    bool valid[2];
    float gx[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        row[g] = r0 + 2 * rho + g;
        valid[g] = row[g] < a.M;
        bidx[g] = valid[g] ? row[g] / a.k : 0;
        gx[g] = valid[g] ? a.gx[row[g]] : 0.0f;
    }
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);

    auto stage = [&](int ng, int buf) {
        char* d = smem + buf * unit;
        stage_image<4>(a.img1 + (size_t)ng * u1, d, u1, wave, lane);
        stage_image<4>(a.img2 + (size_t)ng * 2 * MT2 * 1024, d + u1, 2 * MT2 * 1024, wave, lane);
    };
    auto load_x = [&](int ng, uint4 (&xv)[2][2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                xv[p][g] = make_uint4(0, 0, 0, 0);
                const int fbase = 64 * ng + 32 * p;      // wave-uniform guard; invalid rows read image 0 and are zeroed by gx = 0
                if (fbase < a.Xp32) xv[p][g] = *(const uint4*)(a.XB + (size_t)bidx[g] * a.ldXB + fbase + 8 * q);
            }
    };
    stage(0, 0);

    uint4 bfr[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (KTC ? ks < KTC : ks < KT) {
                v = *(const uint4*)(a.G2 + (size_t)min(row[g], a.M - 1) * a.ldG + ks * 32 + q * 8);
                if (!valid[g]) v = make_uint4(0, 0, 0, 0);
            }
            bfr[ks][g] = v;
        }
    uint4 xv[2][2], xv_n[2][2];
    load_x(0, xv);

    f32x4 acc2[16][2];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc2[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    uint4 stP[2][2];
    int st_ng = -1;
    auto emit_stores = [&]() {
        if (st_ng < 0) return;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int fbase = 64 * st_ng + 32 * p;
            if (fbase < a.Xp32 && a.DLP) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (valid[g]) *(uint4*)(a.DLP + (size_t)row[g] * a.Xp32 + fbase + 8 * q) = stP[p][g];
            }
        }
        st_ng = -1;
    };

    OB_STAMP(0)   // prologue
    for (int ng = 0; ng < a.NG; ++ng) {
        const int buf = ng & 1;
        wait_all_vmem();
        OB_STAMP(1)   // vmcnt wait
        __syncthreads();
        OB_STAMP(2)   // barrier
        if (ng + 1 < a.NG) stage(ng + 1, buf ^ 1);
        emit_stores();
        if (ng + 1 < a.NG) load_x(ng + 1, xv_n);
        OB_STAMP(3)   // issue DMA + stores + x loads
        const char* l1 = smem + buf * unit + a_off;
        const char* lbias = smem + buf * unit + KT * 4096;
        const char* l2 = smem + buf * unit + u1 + a_off;

        f32x4 acc[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (KTC) {
            lds_pipeline<(KTC ? KTC : 1) * 4, 8>(
                [&](int i) { return *(const uint4*)(l1 + i * 1024); },
                [&](int i, const uint4& av) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[i & 3][g] = mfma16(av, bfr[i >> 2][g], acc[i & 3][g]);
                });
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks < KT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint4 av = *(const uint4*)(l1 + (ks * 4 + t) * 1024);
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[t][g] = mfma16(av, bfr[ks][g], acc[t][g]);
                    }
                }
            }
        }
        OB_STAMP(4)   // MFMA phase 1
        float4 bias4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bias4[t] = *(const float4*)(lbias + (16 * t + 4 * q) * 4);
        auto bias_of = [&](int t, int i) { return i == 0 ? bias4[t].x : i == 1 ? bias4[t].y : i == 2 ? bias4[t].z : bias4[t].w; };

        uint4 bf2[2][2];
        auto dl_body = [&](auto masked) {       // wave-uniform branch: only the last pixel group needs masks
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[2][8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float l = acc[2 * p + (j >> 2)][g][j & 3] + bias_of(2 * p + (j >> 2), j & 3);
                        float d = gx[g] * (bf_at(xv[p][g], j) - sigmoid_fast(l));     // d lpxz / d l = x - sigmoid(l)
                        if (decltype(masked)::value) d = (64 * ng + 32 * p + 16 * (j >> 2) + 4 * q + (j & 3) < a.Xdim) ? d : 0.0f;
                        v[g][j] = d;
                    }
                    bf2[p][g] = make_uint4(pack2(v[g][0], v[g][1]), pack2(v[g][2], v[g][3]), pack2(v[g][4], v[g][5]), pack2(v[g][6], v[g][7]));
                    stP[p][g] = bf2[p][g];
                }
            }
        };
        if (64 * ng + 64 <= a.Xdim) dl_body(std::false_type{});
        else dl_body(std::true_type{});
        st_ng = ng;
        OB_STAMP(5)   // epilogue math

        if (KTC) {
            constexpr int MTC = 2 * (KTC ? KTC : 1);
            lds_pipeline<2 * MTC, 8>(
                [&](int i) { return *(const uint4*)(l2 + i * 1024); },
                [&](int i, const uint4& av) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc2[i % MTC][g] = mfma16(av, bf2[i / MTC][g], acc2[i % MTC][g]);
                });
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int mt = 0; mt < 16; ++mt) {
                    if (mt < MT2) {
                        const uint4 av = *(const uint4*)(l2 + (kk * MT2 + mt) * 1024);
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc2[mt][g] = mfma16(av, bf2[kk][g], acc2[mt][g]);
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int g = 0; g < 2; ++g) xv[p][g] = xv_n[p][g];
        OB_STAMP(6)   // MFMA phase 2
    }
    emit_stores();

    // dpre2 = dg2 * (1 - g2^2); the g2 B-operand registers hold exactly the lane's own features
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (KTC ? ks < KTC : ks < KT) {
            float v[2][8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = bf_at(bfr[ks][g], j);
                    v[g][j] = acc2[2 * ks + (j >> 2)][g][j & 3] * (1.0f - y * y);
                }
                if (valid[g])
                    *(uint4*)(a.DPP + (size_t)row[g] * a.ldG + ks * 32 + 8 * q) =
                        make_uint4(pack2(v[g][0], v[g][1]), pack2(v[g][2], v[g][3]), pack2(v[g][4], v[g][5]), pack2(v[g][6], v[g][7]));
            }
        }
    }
    OB_STAMP(7)   // final epilogue
    if (STAMPS && a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i] = tsum[i];
    }
}

// ---------------------------------------------------------------------------------
// out_bwd_pair_kernel: same maths as out_bwd_kernel, restructured so that two waves per SIMD fit.
// 8 waves = 4 PAIRS; both waves of a pair own the same 32 data rows.  Per 64-pixel group wave
// `half` (0/1) of the pair
//   * recomputes the logits of pixels [32*half, 32*half+32) of the group and turns them into dl,
//   * publishes its two bf16 dl fragments (2 x 1 KiB) in LDS and picks up its partner's,
//   * accumulates dg2 for hidden tiles [KTC*half, KTC*half + KTC) over all 64 pixels.
// Registers per wave: 56 (g2) + 16 + 56 (dg2 half) + ~50 instead of ~350, so DMA issue, MFMA and
// the sigmoid epilogue of the two co-resident waves overlap.
// ---------------------------------------------------------------------------------
template <int KTC, int PAIRS, bool STAMPS>     // PAIRS wave pairs per workgroup (32 rows each); 2 waves per SIMD on the CU either way
__global__ __launch_bounds__(PAIRS * 128, 2) void out_bwd_pair_kernel(OutBwdArgs a) {
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int KT = KTC, MH = KTC, NWV = 2 * PAIRS;
    // ONE weight image per pixel group serves both products: the W^T blocks (pixels x hidden) are read row-wise
    // (ds_read_b128) for the logits and column-wise through the hardware-transposing ds_read_b64_tr_b16 for dg2.
    constexpr int u1 = KT * 4096 + 1024;           // W^T group incl. its bias block
    constexpr int unit = u1;
    char* const xch = smem + 2 * unit;             // [pair][half 2][g 2][1 KiB] dl fragments
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, half = wave & 1;
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = (blockIdx.x * PAIRS + pair) * 32;
    int row[2], bidx[2];
    bool valid[2];
    float gx[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        row[g] = r0 + 2 * rho + g;
        valid[g] = row[g] < a.M;
        bidx[g] = valid[g] ? row[g] / a.k : 0;
        gx[g] = valid[g] ? a.gx[row[g]] : 0.0f;
    }
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    // transposing read: lane 4q'+p of quad q supplies the address of pixel row 4q+q', hidden columns 4p..4p+3
    // (an 8-byte piece of the row's 16-byte chunk p); lane i of the quad receives hidden column i for 4 pixels
    const int tr_off = (4 * q + (rho >> 2)) * 64 + (((rho & 3) ^ hperm(q)) * 16);

    // the two weight images of a group are one contiguous LDS unit of NP 1 KiB DMA pieces; wave w moves
    // pieces w, w+8, ...  Piece i of the next group is issued between the MFMAs of the current one.
    constexpr int NP = unit / 1024;
    auto dma_piece = [&](int ng, int buf, int idx) {
        const int p = wave + NWV * idx;               // wave-uniform
        if (p < NP) {
            const char* src = a.img1 + (size_t)ng * u1 + (size_t)p * 1024;
            glds16(src + lane * 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * unit) + (uint32_t)p * 1024u)));
        }
    };
    constexpr int NIDX = (NP + NWV - 1) / NWV;
    auto stage = [&](int ng, int buf) {
#pragma unroll
        for (int idx = 0; idx < NIDX; ++idx) dma_piece(ng, buf, idx);
    };
    auto load_x = [&](int ng, uint4 (&xv)[2]) {
        const int fbase = 64 * ng + 32 * half;      // wave-uniform guard; invalid rows read image 0, zeroed by gx = 0
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            xv[g] = make_uint4(0, 0, 0, 0);
            if (fbase < a.Xp32) xv[g] = *(const uint4*)(a.XB + (size_t)bidx[g] * a.ldXB + fbase + 8 * q);
        }
    };
    stage(0, 0);

    uint4 bfr[KT][2];
#pragma unroll
    for (int ks = 0; ks < KT; ++ks)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 v = *(const uint4*)(a.G2 + (size_t)min(row[g], a.M - 1) * a.ldG + ks * 32 + q * 8);
            if (!valid[g]) v = make_uint4(0, 0, 0, 0);
            bfr[ks][g] = v;
        }
    uint4 xv[2], xv_n[2];
    load_x(0, xv);

    f32x4 acc2[MH][2];
#pragma unroll
    for (int t = 0; t < MH; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc2[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    uint4 stP[2];
    int st_ng = -1;
    auto emit_stores = [&]() {
        if (st_ng < 0) return;
        const int fbase = 64 * st_ng + 32 * half;
        if (fbase < a.Xp32 && a.DLP) {       // dl in P-layout: the lane's 8 features of this 32-pixel step are one 16-byte chunk
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (valid[g]) *(uint4*)(a.DLP + (size_t)row[g] * a.Xp32 + fbase + 8 * q) = stP[g];
        }
        st_ng = -1;
    };

    OB_STAMP(0)
    for (int ng = 0; ng < a.NG; ++ng) {
        const int buf = ng & 1;
        wait_all_vmem();
        OB_STAMP(1)
        __syncthreads();
        OB_STAMP(2)
        const bool more = ng + 1 < a.NG;
        emit_stores();
        if (more) load_x(ng + 1, xv_n);
        OB_STAMP(3)
        const char* l1 = smem + buf * unit + a_off + (2 * half) * 1024;          // this half's two pixel tiles
        const char* lbias = smem + buf * unit + KT * 4096 + (32 * half) * 4;
        const char* l2 = smem + buf * unit + tr_off;                                 // transposed view of the same image

        f32x4 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        lds_pipeline<KT * 2, 8>(
            [&](int i) { return *(const uint4*)(l1 + ((i >> 1) * 4 + (i & 1)) * 1024); },
            [&](int i, const uint4& av) {
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[i & 1][g] = mfma16(av, bfr[i >> 1][g], acc[i & 1][g]);
            },
            [&](int i) { if (more && (i & 1) == 0 && (i >> 1) < NIDX) dma_piece(ng + 1, buf ^ 1, i >> 1); });
        if (more) {
#pragma unroll
            for (int idx = KT; idx < NIDX; ++idx) dma_piece(ng + 1, buf ^ 1, idx);
        }
        OB_STAMP(4)
        float4 bias4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bias4[t] = *(const float4*)(lbias + (16 * t + 4 * q) * 4);
        auto bias_of = [&](int t, int i) { return i == 0 ? bias4[t].x : i == 1 ? bias4[t].y : i == 2 ? bias4[t].z : bias4[t].w; };

        uint4 own[2];
        auto dl_body = [&](auto masked) {       // wave-uniform branch: only the last pixel group needs masks
            float v[2][8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float l = acc[j >> 2][g][j & 3] + bias_of(j >> 2, j & 3);
                    float d = gx[g] * (bf_at(xv[g], j) - sigmoid_fast(l));     // d lpxz / d l = x - sigmoid(l)
                    if (decltype(masked)::value) d = (64 * ng + 32 * half + 16 * (j >> 2) + 4 * q + (j & 3) < a.Xdim) ? d : 0.0f;
                    v[g][j] = d;
                }
                own[g] = make_uint4(pack2(v[g][0], v[g][1]), pack2(v[g][2], v[g][3]), pack2(v[g][4], v[g][5]), pack2(v[g][6], v[g][7]));
            }
            stP[0] = own[0];
            stP[1] = own[1];
        };
        if (64 * ng + 32 * half + 32 <= a.Xdim) dl_body(std::false_type{});
        else dl_body(std::true_type{});
        st_ng = ng;

        OB_STAMP(5)
        // exchange the dl fragments inside the pair
#pragma unroll
        for (int g = 0; g < 2; ++g) *(uint4*)(xch + ((pair * 2 + half) * 2 + g) * 1024 + lane * 16) = own[g];
        __syncthreads();
        uint4 bf2[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint4 other = *(const uint4*)(xch + ((pair * 2 + (half ^ 1)) * 2 + g) * 1024 + lane * 16);
            bf2[0][g] = half ? other : own[g];       // selects, not runtime-indexed arrays (those go to scratch)
            bf2[1][g] = half ? own[g] : other;
        }

        lds_pipeline<2 * MH, 8>(
            [&](int i) {       // A fragment (hidden tile mt, pixel k-step kk) = two transposed 4x16 blocks of pixel tiles 2kk, 2kk+1
                const int kk = i / MH, mt = MH * half + (i % MH);        // mt wave-uniform: hidden k-step mt>>1, half mt&1
                typedef __attribute__((ext_vector_type(4))) short v4s;
                const char* p0 = l2 + ((mt >> 1) * 4 + 2 * kk) * 1024 + 8 * (mt & 1);
                const v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
                const v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p0 + 1024));
                const uint2 lo = __builtin_bit_cast(uint2, r0), hi = __builtin_bit_cast(uint2, r1);
                return make_uint4(lo.x, lo.y, hi.x, hi.y);
            },
            [&](int i, const uint4& av) {
#pragma unroll
                for (int g = 0; g < 2; ++g) acc2[i % MH][g] = mfma16(av, (i / MH) ? bf2[1][g] : bf2[0][g], acc2[i % MH][g]);
            });
#pragma unroll
        for (int g = 0; g < 2; ++g) xv[g] = xv_n[g];
        OB_STAMP(6)
    }
    emit_stores();

    // dpre2 = dg2 * (1 - g2^2) for this half's hidden tiles; tile mt = MH*half + t  <->  (ks = mt>>1, h = mt&1)
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int mt = MH * half + t;      // wave-uniform
        float v[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 gsel = bfr[0][g];
#pragma unroll
            for (int ks = 1; ks < KT; ++ks)
                if ((mt >> 1) == ks) gsel = bfr[ks][g];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y0 = bf_at(gsel, i), y1 = bf_at(gsel, 4 + i);
                const float y = (mt & 1) ? y1 : y0;
                v[g][i] = acc2[t][g][i] * (1.0f - y * y);
            }
            if (valid[g])
                *(uint2*)(a.DPP + (size_t)row[g] * a.ldG + (mt >> 1) * 32 + 8 * q + 4 * (mt & 1)) = make_uint2(pack2(v[g][0], v[g][1]), pack2(v[g][2], v[g][3]));
        }
    }
    OB_STAMP(7)
    if (STAMPS && a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * NWV + wave) * 8 + i] = tsum[i];
    }
}

// ---------------------------------------------------------------------------------
// out_bwd_s_kernel: output-layer backward when the forward pass kept s = x - sigmoid(l) (bf16, P-layout):
//   dg2 = s W^T  (k = pixels; the B operand is s straight from HBM, one 16-byte load per k-step and column group)
//   dpre2 = gx[row] * dg2 * (1 - g2^2)          (the row weight of the objective is applied in fp32, at the end)
// No logits recompute, no sigmoid, no fragment exchange: half the LDS bytes and none of the VALU work of
// out_bwd_pair_kernel.  4 waves x 32 rows; the W^T image streams through LDS one 64-pixel group at a time (same
// image and transposing reads as the pair kernel's second product), DMA pieces issued between the MFMAs.
// ---------------------------------------------------------------------------------
template <int KTC>
__global__ __launch_bounds__(256, 2) void out_bwd_s_kernel(OutBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int KT = KTC, MT = 2 * KTC;
    constexpr int unit = KT * 4096 + 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    int row[2], rowc[2];
    bool valid[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) { row[g] = r0 + 2 * rho + g; valid[g] = row[g] < a.M; rowc[g] = min(row[g], a.M - 1); }
    const int tr_off = (4 * q + (rho >> 2)) * 64 + (((rho & 3) ^ hperm(q)) * 16);

    constexpr int NP = unit / 1024, NIDX = (NP + 3) / 4;
    auto dma_piece = [&](int ng, int buf, int idx) {
        const int p = wave + 4 * idx;                 // wave-uniform
        if (p < NP)
            glds16(a.img1 + (size_t)ng * unit + (size_t)p * 1024 + lane * 16,
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + buf * unit) + (uint32_t)p * 1024u)));
    };
    auto load_s = [&](int ng, uint4 (&sf)[2][2]) {   // clamped rows, zeroed by a select: no branch per load
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                sf[kk][g] = make_uint4(0, 0, 0, 0);
                const int fbase = 64 * ng + 32 * kk;      // wave-uniform guard
                if (fbase < a.Xp32) {
                    const uint4 v = *(const uint4*)(a.SP + (size_t)rowc[g] * a.Xp32 + fbase + 8 * q);
                    sf[kk][g] = valid[g] ? v : make_uint4(0, 0, 0, 0);
                }
            }
    };
    // small row counts: blockIdx.y takes a slice of the pixel groups and leaves its fp32 partial sums in a.part
    const int ng0 = a.part ? (int)blockIdx.y * a.gpb : 0;
    const int ng1 = a.part ? min(a.NG, ng0 + a.gpb) : a.NG;
#pragma unroll
    for (int idx = 0; idx < NIDX; ++idx) dma_piece(ng0, ng0 & 1, idx);
    uint4 sf[2][2], sf_n[2][2];
    load_s(ng0, sf);

    f32x4 acc2[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc2[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    for (int ng = ng0; ng < ng1; ++ng) {
        const int buf = ng & 1;
        wait_all_vmem();
        __syncthreads();
        const bool more = ng + 1 < ng1;
        if (more) load_s(ng + 1, sf_n);
        const char* l2 = smem + buf * unit + tr_off;
        lds_pipeline<2 * MT, 8>(
            [&](int i) {       // A fragment (hidden tile mt, pixel k-step kk) = two transposed 4x16 blocks of pixel tiles 2kk, 2kk+1
                const int kk = i / MT, mt = i % MT;
                typedef __attribute__((ext_vector_type(4))) short v4s;
                const char* p0 = l2 + ((mt >> 1) * 4 + 2 * kk) * 1024 + 8 * (mt & 1);
                const v4s t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
                const v4s t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p0 + 1024));
                const uint2 lo = __builtin_bit_cast(uint2, t0), hi = __builtin_bit_cast(uint2, t1);
                return make_uint4(lo.x, lo.y, hi.x, hi.y);
            },
            [&](int i, const uint4& av) {
#pragma unroll
                for (int g = 0; g < 2; ++g) acc2[i % MT][g] = mfma16(av, (i / MT) ? sf[1][g] : sf[0][g], acc2[i % MT][g]);
            },
            [&](int i) { if (more && (i & 1) == 0 && (i >> 1) < NIDX) dma_piece(ng + 1, buf ^ 1, i >> 1); });
        if (more) {
#pragma unroll
            for (int idx = MT; idx < NIDX; ++idx) dma_piece(ng + 1, buf ^ 1, idx);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int g = 0; g < 2; ++g) sf[kk][g] = sf_n[kk][g];
    }

    if (a.part) {      // partial dg2 in P order (feature position ks*32 + 8q + j), finished by out_bwd_finish_kernel
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (valid[g]) {
                    float* dst = a.part + ((size_t)blockIdx.y * a.M + row[g]) * a.ldG + ks * 32 + 8 * q;
                    *(float4*)dst = make_float4(acc2[2 * ks][g][0], acc2[2 * ks][g][1], acc2[2 * ks][g][2], acc2[2 * ks][g][3]);
                    *(float4*)(dst + 4) = make_float4(acc2[2 * ks + 1][g][0], acc2[2 * ks + 1][g][1], acc2[2 * ks + 1][g][2], acc2[2 * ks + 1][g][3]);
                }
        return;
    }
    // dpre2 = gx * dg2 * (1 - g2^2); the lane's 8 features of hidden k-step ks are tiles 2ks (j < 4) and 2ks+1 (j >= 4)
    float gxv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) gxv[g] = valid[g] ? a.gx[row[g]] : 0.0f;
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint4 y8 = *(const uint4*)(a.G2 + (size_t)rowc[g] * a.ldG + ks * 32 + q * 8);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = bf_at(y8, j);
                v[j] = gxv[g] * acc2[2 * ks + (j >> 2)][g][j & 3] * (1.0f - y * y);
            }
            if (valid[g])
                *(uint4*)(a.DPP + (size_t)row[g] * a.ldG + ks * 32 + 8 * q) =
                    make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
        }
    }
}

// ---------------------------------------------------------------------------------
// dec_bwd_kernel: the decoder's whole dX chain in ONE launch (large row counts, s kept by the forward pass):
//   dg2 = s W3^T ; dpre2 = g_r * dg2 * (1 - g2^2)            (out_bwd_s_kernel's product, iwae1.py:74-75,111)
//   dpre1 = (dpre2 V2^T) * (1 - g1^2)                         (dense_kernel<EPI_DX>'s)
//   dz    = dpre1 V1^T                                        (dense_kernel<EPI_F32>'s)
// A converted accumulator IS the next product's B operand (layout.h), so dpre2 and dpre1 go from product to product in
// registers; they are stored once (the hidden layers' weight gradients read them) and never read back here.  As three
// launches the chain was 40 + 29 + 18 us with a boundary, a tail and a 22 MB re-read between each pair; the weights of all
// three products stream through the same two LDS buffers as one sequence of 29 KiB units (13 pixel groups of W3^T read
// through the transposing ds_read_b64_tr_b16, then the 4 + 2 out-feature groups of the backward images of V2 and V1).
// 4 waves x 32 rows, two workgroups per CU (<= 256 registers): dg2 accumulators 112, then dpre2 / dpre1 fragments 56 each.
// ---------------------------------------------------------------------------------
// <KTC, NW, G>: NW waves per workgroup, G row groups of 16 rows per wave.  <.., 4, 2> (round 2): 4 waves x 32 rows, two workgroups per CU = 2 waves per SIMD at <= 256
// registers.  <.., 8, 1> (round 4, DESIGN item 69): 8 waves x 16 rows, two workgroups per CU = 4 waves per SIMD at <= 128 registers -- the same 128 rows per weight unit,
// half the accumulators per wave, twice the waves to hide each other's latencies (the kernel's units take 3.8 k cycles for 0.9 k cycles of MFMAs per wave).
template <int KTC, int NW = 4, int G = 2>
__global__ __launch_bounds__(64 * NW, 2) void dec_bwd_kernel(DecBwdArgs d) {
#ifdef IWAE_DENSE_STAMPS       // diagnostic build: cycles per phase and wave -> d.o.stamps[wave][8] (p1 wait | p1 multiply | dpre2 | p2/p3 wait | p2/p3 multiply + epilogue | - | end)
    unsigned long long ds_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ds_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ds_prev)::"memory");
#endif
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const OutBwdArgs& a = d.o;
    constexpr int KT = KTC, MT = 2 * KTC;
    constexpr int unit = KT * 4096 + 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int r0 = (blockIdx.x * NW + wave) * (16 * G);
    int row[G], rowc[G];
    bool valid[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { row[g] = r0 + G * rho + g; valid[g] = row[g] < a.M; rowc[g] = min(row[g], a.M - 1); }
    const int a_off = rho * 64 + ((q ^ hperm(rho >> 2)) * 16);
    constexpr int MG2 = (KTC + 1) / 2;                             // 64-feature groups of the hidden width (32*KTC features)
    const int U1 = a.NG, U2 = U1 + MG2, U3 = U2 + d.MG1;          // unit ranges of the three products

    constexpr int NP = unit / 1024, NIDX = (NP + NW - 1) / NW;
    auto dma_piece = [&](int u, int idx) {
        const int p = wave + NW * idx;                // wave-uniform
        if (p >= NP) return;
        // product 1 reads W3 from its K-MAJOR image (blocks [pixel k-step][hidden tile]: a pixel group is 2 * MT contiguous 1 KiB blocks, each
        // ALREADY the A fragment of dg2 = s W3^T): plain conflict-free ds_read_b128, where the MG-major image of W3^T needed two transposing
        // ds_read_b64_tr_b16 per fragment with their inherent 2-way bank conflict (41 % of this kernel's LDS cycles)
        if (u < U1 && p >= 2 * MT) return;
        const char* src = u < U1 ? a.img2 + (size_t)u * (2 * MT * 1024) : u < U2 ? d.imgB2 + (size_t)(u - U1) * unit : d.imgB1 + (size_t)(u - U2) * unit;
        glds16(src + (size_t)p * 1024 + lane * 16,
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + (u & 1) * unit) + (uint32_t)p * 1024u)));
    };
    auto load_s = [&](int ng, uint4 (&sf)[2][G]) {   // clamped rows, zeroed by a select: no branch per load
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                sf[kk][g] = make_uint4(0, 0, 0, 0);
                const int fbase = 64 * ng + 32 * kk;      // wave-uniform guard
                if (fbase < a.Xp32) {
                    const uint4 v = *(const uint4*)(a.SP + (size_t)(WG_DBG(a, 32) ? (rowc[g] & 31) : rowc[g]) * a.Xp32 + fbase + 8 * q);
                    sf[kk][g] = valid[g] ? v : make_uint4(0, 0, 0, 0);
                }
            }
    };
#pragma unroll
    for (int idx = 0; idx < NIDX; ++idx) dma_piece(0, idx);
    uint4 d2f[KT][G], d1f[KT][G];
    {
        // ---------------- product 1: dg2 = s W3^T over the pixel groups
        uint4 sf[2][G], sf_n[2][G];
        load_s(0, sf);
        f32x4 acc2[MT][G];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int g = 0; g < G; ++g) acc2[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        for (int ng = 0; ng < U1; ++ng) {
            const int buf = ng & 1;
            DS_STAMP(1);
            wait_all_vmem();
            __syncthreads();
            DS_STAMP(0);
            if (ng + 1 < U1) load_s(ng + 1, sf_n);
            const char* l2 = smem + buf * unit + a_off;
            lds_pipeline<2 * MT, 8>(
                [&](int i) { return *(const uint4*)(l2 + i * 1024); },      // A fragment (pixel k-step kk = i / MT, hidden tile mt = i % MT): block kk * MT + mt of the unit
                [&](int i, const uint4& av) {
#pragma unroll
                    for (int g = 0; g < G; ++g) acc2[i % MT][g] = mfma16(av, (i / MT) ? sf[1][g] : sf[0][g], acc2[i % MT][g]);
                },
                [&](int i) { if ((i & 1) == 0 && (i >> 1) < NIDX) dma_piece(ng + 1, i >> 1); });      // unit U1 (first group of V2's image) follows the last pixel group
#pragma unroll
            for (int idx = MT; idx < NIDX; ++idx) dma_piece(ng + 1, idx);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int g = 0; g < G; ++g) sf[kk][g] = sf_n[kk][g];
        }
        // dpre2 = gx * dg2 * (1 - g2^2); the lane's 8 features of hidden k-step ks are tiles 2ks (j < 4) and 2ks+1 (j >= 4)
        DS_STAMP(1);
        float gxv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) gxv[g] = valid[g] ? a.gx[row[g]] : 0.0f;
        // the stored g2 of the wave's rows: ALL fragments requested before the first is used (one round trip; as a load per use the
        // 14 of them came back one after the other: 18.6k of the kernel's 107k cycles per wave)
        uint4 y8a[KT][G];
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int g = 0; g < G; ++g) y8a[ks][g] = *(const uint4*)(a.G2 + (size_t)rowc[g] * a.ldG + ks * 32 + q * 8);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int g = 0; g < G; ++g) asm volatile("" : "+v"(y8a[ks][g].x), "+v"(y8a[ks][g].y), "+v"(y8a[ks][g].z), "+v"(y8a[ks][g].w));
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint4 y8 = y8a[ks][g];
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = bf_at(y8, j);
                    v[j] = gxv[g] * acc2[2 * ks + (j >> 2)][g][j & 3] * (1.0f - y * y);
                }
                const uint4 frag = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                d2f[ks][g] = valid[g] ? frag : make_uint4(0, 0, 0, 0);
                if (valid[g]) *(uint4*)(a.DPP + (size_t)row[g] * a.ldG + ks * 32 + 8 * q) = frag;
            }
        }
    }
    DS_STAMP(2);
    // ---------------- products 2 and 3: Y^T = backward image x X^T with X in registers, one 64-out-feature group per unit
    constexpr int NF = KT * 4, STEP = (NF / NIDX > 0) ? NF / NIDX : 1;
    auto group_mfma = [&](int u, const uint4 (&xin)[KT][G], f32x4 (&acc)[4][G]) {
        const int buf = u & 1;
        DS_STAMP(4);
        wait_all_vmem();
        __syncthreads();
        DS_STAMP(3);
        const bool more = u + 1 < U3;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[t][g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        const char* lb = smem + buf * unit + a_off;
        lds_pipeline<NF, 8>(
            [&](int i) { return *(const uint4*)(lb + i * 1024); },
            [&](int i, const uint4& av) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[i & 3][g] = mfma16(av, xin[i >> 2][g], acc[i & 3][g]);
            },
            [&](int i) { if (more && i % STEP == 0 && i / STEP < NIDX) dma_piece(u + 1, i / STEP); });
        if (more) {
#pragma unroll
            for (int idx = (NF + STEP - 1) / STEP; idx < NIDX; ++idx) dma_piece(u + 1, idx);
        }
    };
#pragma unroll
    for (int ks = 0; ks < KT; ++ks)
#pragma unroll
        for (int g = 0; g < G; ++g) d1f[ks][g] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int mg = 0; mg < MG2; ++mg) {
        // stored g1 of this group's 64 features: requested before the MFMAs, used after them
        uint4 y8[2][G];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                y8[p][g] = make_uint4(0, 0, 0, 0);
                if (2 * mg + p < KT) y8[p][g] = *(const uint4*)(d.G1 + (size_t)rowc[g] * a.ldG + (2 * mg + p) * 32 + q * 8);
            }
        f32x4 acc[4][G];
        group_mfma(U1 + mg, d2f, acc);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int kso = 2 * mg + p;
            if (kso < KT) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = bf_at(y8[p][g], j);
                        v[j] = acc[2 * p + (j >> 2)][g][j & 3] * (1.0f - y * y);
                    }
                    const uint4 frag = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                    d1f[kso][g] = valid[g] ? frag : make_uint4(0, 0, 0, 0);       // (kso is a compile-time value: the group loop is unrolled)
                    if (valid[g]) *(uint4*)(d.D1P + (size_t)row[g] * a.ldG + kso * 32 + 8 * q) = frag;
                }
            }
        }
    }
    for (int mg = 0; mg < d.MG1; ++mg) {
        f32x4 acc[4][G];
        group_mfma(U2 + mg, d1f, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int f0 = 64 * mg + 16 * t + 4 * q;
            if (64 * mg + 16 * t < d.ldDZ) {       // wave-uniform
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (valid[g] && d.DZH) *(uint2*)(d.DZH + (size_t)row[g] * d.ldDZ + f0) = make_uint2(pack2(acc[t][g][0], acc[t][g][1]), pack2(acc[t][g][2], acc[t][g][3]));
                    else if (valid[g]) *(float4*)(d.DZ + (size_t)row[g] * d.ldDZ + f0) = make_float4(acc[t][g][0], acc[t][g][1], acc[t][g][2], acc[t][g][3]);
                }
            }
        }
    }
#ifdef IWAE_DENSE_STAMPS
    DS_STAMP(4);
    wait_all_vmem();
    DS_STAMP(6);
    if (a.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) a.stamps[((size_t)blockIdx.x * NW + wave) * 8 + i] = ds_sum[i];
#endif
}

// dpre2 = gx * (sum of the partial dg2 slices) * (1 - g2^2), elementwise in P order: one thread per 8-feature chunk
__global__ __launch_bounds__(256) void out_bwd_finish_kernel(OutBwdArgs a, int nparts) {
    const int nch = a.ldG / 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)a.M * nch) return;
    const int row = (int)(idx / nch), c = (int)(idx - (size_t)row * nch);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int y = 0; y < nparts; ++y) {
        const float* src = a.part + ((size_t)y * a.M + row) * a.ldG + 8 * c;
        const float4 p0 = *(const float4*)src, p1 = *(const float4*)(src + 4);
        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
    }
    const uint4 y8 = *(const uint4*)(a.G2 + (size_t)row * a.ldG + 8 * c);
    const float gx = a.gx[row];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float y = bf_at(y8, j); v[j] = gx * v[j] * (1.0f - y * y); }
    *(uint4*)(a.DPP + (size_t)row * a.ldG + 8 * c) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// ---------------------------------------------------------------------------------
// wgradp_kernel: weight gradient straight from the ROW-major P-layout activations (no feature-major copies):
//   out[i][j] = sum_r X[r][i] * G[r][j],   X: bf16 [rows][ldX], G: bf16 [rows][ldG]  (both P-layout)
// The contraction index (data row) is the strided one in memory, so both MFMA operands are produced by the
// hardware-transposing LDS read: per 64-row chunk the X tile (<=256 features) and the G strip (NW*16
// features) are DMA'd row-major into LDS (16-byte slot s of row r lands at slot s ^ ((r&3)<<2), the swizzle
// is applied on the per-lane SOURCE address), and a fragment is two ds_read_b64_tr_b16 (4 rows x 16 features
// each; 2-way bank conflict by construction: every lane uses one half of a 16-byte slot).  P-layout's feature
// permutation is undone for free by which 8-byte piece a lane addresses.  Rows >= M read a zero line.
// grid = (j-blocks of NW*16 features, i-blocks of 256 features, row splits); fp32 slabs as before.
// ---------------------------------------------------------------------------------
// Register blocking: the NW waves form a WI x (NW/WI) grid; a wave owns 16/WI i-tiles x WI j-tiles (16 accumulator tiles
// either way) and reads 16/WI + WI fragments per 32-row step instead of 16 + 1 -- the transposing LDS reads (2-way bank
// conflict by construction) were what a 64-row chunk waited for (4 352 LDS cycles per chunk against 2 048 MFMA cycles per
// SIMD at 16 x 1).  The row-weighted variant scales every G fragment it reads, so it takes the 8 x 2 shape.
// counted wait: all but the wave's n youngest vector-memory operations (here: LDS-DMA pieces, issued in stage order) are done
__device__ __forceinline__ void wait_vmem_but(int n) {      // n wave-uniform, 0..12
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

// Staging: the operands move through a RING of WG_NST LDS stages of WG_SR = 32 data rows each; while stage c is multiplied the
// DMA of stage c + 3 is issued (pieces spread between the MFMAs), and the top of stage c + 1 waits only for ITS pieces
// (s_waitcnt vmcnt(n) with the two younger stages' pieces still in flight).  Round 1 double-buffered 64-row chunks and
// waited for chunk c + 1 right after issuing its last piece: every chunk exposed one full HBM round trip (~1.5-2 us under
// load against ~1 us of MFMA + LDS work; 3.9 us per chunk measured on the output layer's gradient = 0.13 of the HBM peak).
#define WG_SR 32
#define WG_NST 4
// diagnostic ablations of the weight-gradient kernels (WgradPArgs.dbg: timing only, results are wrong) exist in DIAG=1 builds only;
// in the shipped kernels the conditions fold to false at compile time
// (WG_DBG: defined at the top of the file)
// Shapes <NW waves, IGC i-groups, AI x BJ accumulator tiles per wave>: the waves form an IGC x (NW/IGC) grid, the workgroup's
// output tile is IGC*AI i-tiles x (NW/IGC)*BJ j-tiles.  A 32-row stage costs a wave AI + BJ fragment reads for AI*BJ MFMAs:
//   <16, 4, 4, 4>  256 x 256 features, 0.50 reads per MFMA (round 1's shape; the hidden layers at large row counts)
//   <16, 2, 8, 2>  the same tile for the row-weighted variant (scales its 2 G fragments), 0.63
//   < 8, 2, 7, 4>  224 x 256 features with EIGHT waves (two per SIMD, 28 accumulator tiles each): 0.39 reads per MFMA -- below the
//                  0.5 at which the LDS (2-way conflict of the transposing reads: 128 B/clk) keeps up with the MFMA pipes -- and
//                  half the waves at every barrier; for layers whose input is <= 224 features wide (the decoder's)
//   < 8, 4, 4, 4>  256 x 128 features, 8 waves: small row counts (grouped launch for an encoder block)
template <int NW, int IGC, int AI, int BJ, bool SC>     // SC: G rows carry a per-row weight (a.rowscale), staged through LDS with the tiles
__device__ __forceinline__ void wgradp_body(const WgradPArgs& a, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int XT_BYTES = WG_SR * 512;              // X tile: 32 rows x 512 B (256 features)
    constexpr int JGC = NW / IGC, STRIP = JGC * BJ;    // j-tiles of the workgroup's G strip
    constexpr int GROW = STRIP * 32;                   // G strip row bytes (512 for 16 j-tiles, 256 for 8)
    constexpr int GT_BYTES = WG_SR * GROW;
    constexpr int BUF = XT_BYTES + GT_BYTES + (SC ? 1024 : 0);
    constexpr int XP = XT_BYTES / 1024, GP = GT_BYTES / 1024, NPC = XP + GP + (SC ? 1 : 0), NIDX = (NPC + NW - 1) / NW;
    typedef __attribute__((ext_vector_type(4))) short v4s;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15, qp = l16 >> 2, p = l16 & 3;
    const int ig = wave % IGC, jg = wave / IGC;        // the wave's i-tiles ig*AI .. +AI-1, local j-tiles jg*BJ .. +BJ-1
    const int it0 = by * 16;
    const int nit = min(IGC * AI, a.IT - it0);
    const int split = bz;
    const int rbeg = split * a.rows_per_split;
    const int rend = min(a.M, rbeg + a.rows_per_split);
    const int nstage = (rend - rbeg + WG_SR - 1) / WG_SR;
    const int xcol0 = it0 * 16, gcol0 = bx * STRIP * 16;    // first feature (= P position, both multiples of 32) of the tiles
    int my_pieces = 0;                                       // DMA pieces this wave issues per stage (wave-uniform)
#pragma unroll
    for (int idx = 0; idx < NIDX; ++idx) my_pieces += (wave + NW * idx < NPC) ? 1 : 0;

    // one DMA piece = 1 KiB of LDS = 2 X rows (32 slots each) or 1024/GROW G rows.  A lane fetches the same (row in stage,
    // 16-byte chunk) of its pieces in every stage: the source address is a per-lane base + stage * (32 rows of bytes), both
    // precomputed (as per-stage 64-bit index arithmetic it was ~25 vector instructions per piece, 96 per wave and stage in
    // all against 16 MFMAs -- SQ_INSTS_VALU of the round-2 profile).
    const char* zsrc = a.zero + (lane & 31) * 16;
    const char* pbase[NIDX];       // lane's source at stage 0
    int plim[NIDX];                // the piece's row is valid in stage c iff c * WG_SR < plim
    int pstride[NIDX];             // bytes per stage (wave-uniform)
#pragma unroll
    for (int idx = 0; idx < NIDX; ++idx) {
        const int pc = wave + NW * idx;               // wave-uniform
        pbase[idx] = zsrc; plim[idx] = -(1 << 30); pstride[idx] = 0;
        if (pc < XP) {
            const int rl = 2 * pc + (lane >> 5), sl = lane & 31;
            const int col = xcol0 + (sl ^ ((rl & 3) << 2)) * 8;           // source 16-byte chunk of this LDS slot
            pbase[idx] = (const char*)a.X + ((size_t)(rbeg + rl) * a.ldX + col) * 2;
            plim[idx] = col < a.ldX ? rend - rbeg - rl : -(1 << 30);
            pstride[idx] = WG_SR * a.ldX * 2;
        } else if (SC && pc == XP + GP) {                                // the stage's 32 row weights (128 B; the pad rows of gx are finite)
            pbase[idx] = (const char*)(a.rowscale + rbeg) + (lane & 7) * 16;
            plim[idx] = lane < WG_SR / 4 ? (1 << 30) : -(1 << 30);
            pstride[idx] = WG_SR * 4;
        } else if (pc < NPC) {
            constexpr int SPR = GROW / 16, RPP = 1024 / GROW;            // slots per row, rows per piece
            const int rl = RPP * (pc - XP) + lane / SPR, sl = lane % SPR;
            const int col = gcol0 + (sl ^ ((rl & 3) << 2)) * 8;
            pbase[idx] = (const char*)a.G + ((size_t)(rbeg + rl) * a.ldG + col) * 2;
            plim[idx] = col < a.ldG ? rend - rbeg - rl : -(1 << 30);
            pstride[idx] = WG_SR * a.ldG * 2;
        }
    }
    auto dma_piece = [&](int c, int idx) {
        const int pc = wave + NW * idx;               // wave-uniform
        if (pc >= NPC) return;
        const char* src = (c * WG_SR < plim[idx]) ? pbase[idx] + (size_t)c * (size_t)pstride[idx] : zsrc;
        glds16(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + (c % WG_NST) * BUF) + (uint32_t)pc * 1024u)));
    };

    f32x4 acc[AI][BJ];
#pragma unroll
    for (int t = 0; t < AI; ++t)
#pragma unroll
        for (int u = 0; u < BJ; ++u) acc[t][u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float bsum[BJ];
#pragma unroll
    for (int u = 0; u < BJ; ++u) bsum[u] = 0.0f;
    for (int c = 0; c < min(nstage, WG_NST - 1); ++c) {
#pragma unroll
        for (int idx = 0; idx < NIDX; ++idx) dma_piece(c, idx);
    }
    // lane-constant parts of the transposing-read addresses: row 4q+q' of a 16-row half-step, piece p
    const int xrow_off = (4 * q + qp) * 512, grow_off = (4 * q + qp) * GROW;

    for (int c = 0; c < nstage; ++c) {
        const int buf = c % WG_NST;
        wait_vmem_but(my_pieces * min(WG_NST - 2, nstage - 1 - c));      // stage c has landed; the (<= 2) younger stages may still fly
        __syncthreads();                                                 // ... for every wave; and everyone is done reading stage c - 1
        const bool more = c + WG_NST - 1 < nstage && !WG_DBG(a, 1);       // stage c + 3 goes into the buffer stage c - 1 has just left
        int dma_idx = 0;
        auto dma_next = [&]() {
            if (more && dma_idx < NIDX) dma_piece(c + WG_NST - 1, dma_idx);
            ++dma_idx;
        };
        const char* xb = smem + buf * BUF + xrow_off + p * 16;
        const char* gbase = smem + buf * BUF + XT_BYTES + grow_off;
        {
            // B fragments: 8 data rows (two 4-row blocks) of each of this wave's BJ j-tiles (local tile jl -> chunk 4*(jl>>1)+p, half jl&1)
            uint4 g[BJ];
#pragma unroll
            for (int u = 0; u < BJ; ++u) {
                const int jl = jg * BJ + u;
                const char* gb = gbase + ((((4 * (jl >> 1)) ^ (qp << 2)) + p) * 16) + 8 * (jl & 1);
                const v4s g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)gb);
                const v4s g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(gb + 16 * GROW));
                const uint2 glo = __builtin_bit_cast(uint2, g0), ghi = __builtin_bit_cast(uint2, g1);
                g[u] = make_uint4(glo.x, glo.y, ghi.x, ghi.y);
            }
            if (SC && !WG_DBG(a, 8)) {      // the lane's 8 data rows: 4q..4q+3 and 16+4q..16+4q+3 of this 32-row stage (output layer: G = s, weight = dLoss/dlpxz)
                const float* scl = (const float*)(smem + buf * BUF + XT_BYTES + GT_BYTES) + 4 * q;
                const float4 s0 = *(const float4*)scl, s1 = *(const float4*)(scl + 16);
#pragma unroll
                for (int u = 0; u < BJ; ++u)
                    g[u] = make_uint4(pack2(bflo(g[u].x) * s0.x, bfhi(g[u].x) * s0.y), pack2(bflo(g[u].y) * s0.z, bfhi(g[u].y) * s0.w),
                                      pack2(bflo(g[u].z) * s1.x, bfhi(g[u].z) * s1.y), pack2(bflo(g[u].w) * s1.z, bfhi(g[u].w) * s1.w));
            }
            if (ig == 0 && !WG_DBG(a, 4)) {       // bias gradient = column sums of G: one wave per j-tile (wave-uniform branch)
#pragma unroll
                for (int u = 0; u < BJ; ++u)
                    bsum[u] += bflo(g[u].x) + bfhi(g[u].x) + bflo(g[u].y) + bfhi(g[u].y) + bflo(g[u].z) + bfhi(g[u].z) + bflo(g[u].w) + bfhi(g[u].w);
            }
            if (!WG_DBG(a, 2)) lds_pipeline<AI, (AI < 3 ? AI : 3)>(
                [&](int t) {       // A fragment of i-tile i: P chunk 4*(i>>1)+p (swizzled by the row), half i&1
                    const int i = ig * AI + t;
                    const char* p0 = xb + ((((4 * (i >> 1)) ^ (qp << 2))) * 16) + 8 * (i & 1);
                    const v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
                    const v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p0 + 16 * 512));
                    const uint2 lo = __builtin_bit_cast(uint2, r0), hi = __builtin_bit_cast(uint2, r1);
                    return make_uint4(lo.x, lo.y, hi.x, hi.y);
                },
                [&](int t, const uint4& av) {
                    if (ig * AI + t < nit) {
#pragma unroll
                        for (int u = 0; u < BJ; ++u) acc[t][u] = mfma16(av, g[u], acc[t][u]);
                    }
                },
                [&](int t) { if (t == 0 || (AI > 1 && t == AI / 2) || (AI > 3 && t == AI - 1)) dma_next(); });
        }
        while (dma_idx < NIDX) dma_next();
    }

    // D: lane(col j = lane&15, quad q) reg ii -> out[i = 16*it + 4q + ii][j]
    float* slab = a.slabW + (size_t)split * a.IT * 16 * a.JT * 16;
    if (WG_DBG(a, 16)) return;
#pragma unroll
    for (int u = 0; u < BJ; ++u) {
        const int jt = bx * STRIP + jg * BJ + u;
        if (jt < a.JT) {
#pragma unroll
            for (int t = 0; t < AI; ++t) {
                const int it = ig * AI + t;
                if (it < nit) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        slab[(size_t)((it0 + it) * 16 + 4 * q + ii) * (a.JT * 16) + jt * 16 + l16] = acc[t][u][ii];
                }
            }
            if (by == 0 && ig == 0) {
                float v = bsum[u];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (q == 0) a.slabB[(size_t)split * a.JT * 16 + jt * 16 + l16] = v;
            }
        }
    }
}

template <int NW, int IGC, int AI, int BJ, bool SC>
__global__ __launch_bounds__(NW * 64, (AI * BJ > 16) ? 2 : NW / 4) void wgradp_kernel(WgradPArgs a) {
    // XCD-aware block order: the hardware deals workgroups round-robin to the 8 XCDs (private L2 each).  Renumber
    // so that blocks which share operands -- the j-blocks / i-blocks of one row split -- sit on ONE XCD and are
    // dispatched back to back: the shared X tile then comes from HBM once instead of once per j-block.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int gx = gridDim.x, gy = gridDim.y, nb = gx * gy * (int)gridDim.z;
    if ((nb & 7) == 0) {
        const int L = bx + gx * (by + gy * bz);
        const int V = (L & 7) * (nb >> 3) + (L >> 3);
        bx = V % gx; by = (V / gx) % gy; bz = V / (gx * gy);
    }
    wgradp_body<NW, IGC, AI, BJ, SC>(a, bx, by, bz);
}

// ---------------------------------------------------------------------------------
// wgradws_kernel: the weight gradient of a layer whose input is <= 224 features wide (the decoder's three layers) at large
// row counts, with SPECIALISED waves.  Ablations of wgradp_kernel on the output layer (alone on the machine, 90 us): without its
// MFMAs and A-fragment reads 92 us, without the DMA 73, without the row scaling 74, without the bias sums 82, with none of the
// four 29 -- every wave's own instruction stream (DMA issue ~14 instructions per 1 KiB piece, 80 vector instructions of row
// scaling, 64 of bias sums, ~100 scalar ones of loop skeleton, per 28 MFMAs) is what a 32-row stage takes, run in sequence by
// waves that a barrier keeps in lockstep; the MFMA pipe idles 3/4 of the time.  Here:
//   * 8 COMPUTE waves (2 x 4 grid, 7 x 4 accumulator tiles each = 224 x 256 features) only read fragments and multiply; the bias
//     gradient (column sums of G) comes out of the matrix pipe too (one MFMA per G fragment against an all-ones A fragment);
//   * 4 LOADER waves issue all LDS-DMA of the stage ring (3 stages ahead); in the row-weighted variant (the output layer: G = the
//     stored s, weight = dLoss/dlpxz of the row) the G strip travels through the loaders' REGISTERS instead and is multiplied by
//     the row weight on the way into LDS (bf16, one rounding: dl = bf16(g_r * s)) -- see the loader branch below.
// One barrier per stage: behind it the compute waves own stage c + 1 (landed, weighted) and the loaders own the buffer of
// stage c (to refill).
// ---------------------------------------------------------------------------------
// counted wait for any count 0..24 (wave-uniform)
__device__ __forceinline__ void wait_vmem_but_ws(int n) {
#define IWAE_WS_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        IWAE_WS_CASE(0) IWAE_WS_CASE(1) IWAE_WS_CASE(2) IWAE_WS_CASE(3) IWAE_WS_CASE(4) IWAE_WS_CASE(5) IWAE_WS_CASE(6) IWAE_WS_CASE(7)
        IWAE_WS_CASE(8) IWAE_WS_CASE(9) IWAE_WS_CASE(10) IWAE_WS_CASE(11) IWAE_WS_CASE(12) IWAE_WS_CASE(13) IWAE_WS_CASE(14) IWAE_WS_CASE(15)
        IWAE_WS_CASE(16) IWAE_WS_CASE(17) IWAE_WS_CASE(18) IWAE_WS_CASE(19) IWAE_WS_CASE(20) IWAE_WS_CASE(21) IWAE_WS_CASE(22) IWAE_WS_CASE(23)
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
#undef IWAE_WS_CASE
}
// <SC, BJ, NLW>: BJ = j-tiles per compute wave (the workgroup's G strip is 4*BJ tiles wide), NLW = loader waves.
//   <.., 4, 4>: 224 x 256 features per workgroup, 8 + 4 waves at <= 168 registers (three per SIMD)
//   <.., 2, 8>: 224 x 128 features, 8 + 8 waves at <= 128 registers (four per SIMD): a third of the DMA issues and a quarter of the
//               scaling per loader and stage -- the loaders' instruction stream is what a stage takes (see above) -- for 1.75x the
//               X-tile traffic out of L2 (seven column blocks instead of four)
#ifdef IWAE_DENSE_STAMPS      // diagnostic build: cycles per phase and wave -> a.stamps[workgroup * waves + wave][8]
#define WS_STAMP(slot)                                                                 \
    {                                                                                  \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        ws_sum[slot] += t_ - ws_prev;                                                  \
        ws_prev = t_;                                                                  \
    }
#define WS_STAMP_OUT()                                                                                                         \
    if (a.stamps && lane == 0) {                                                                                               \
        for (int i_ = 0; i_ < 8; ++i_) a.stamps[((size_t)sid * (NCW + NLW) + wave) * 8 + i_] = ws_sum[i_];                     \
    }
#else
#define WS_STAMP(slot)
#define WS_STAMP_OUT()
#endif
// XCD-aware block order (see wgradp_kernel): the j-blocks of one row split share an XCD's L2
__device__ __forceinline__ void wgradws_block(int& bx, int& bz, const int gx, const int nz) {
    const int nb = gx * nz;
    if ((nb & 7) == 0) {
        const int L = bx + gx * bz;
        const int V = (L & 7) * (nb >> 3) + (L >> 3);
        bx = V % gx; bz = V / gx;
    }
}
// jt0: first j-tile (16 out-features) of the workgroup's G strip, bz: its row split, sid: its slot in the diagnostic stamp buffer
template <bool SC, int BJ, int NLW>
__device__ __forceinline__ void wgradws_body(const WgradPArgs& a, const int jt0, const int bz, const int sid) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
#ifdef IWAE_DENSE_STAMPS
    unsigned long long ws_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ws_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ws_prev)::"memory");
#endif
    constexpr int AI = 7, IGC = 2, NCW = 8, STRIP = 4 * BJ;
    constexpr int XT_BYTES = WG_SR * 512, GROW = STRIP * 32, GT_BYTES = WG_SR * GROW, BUF = XT_BYTES + GT_BYTES;
    constexpr int XP = XT_BYTES / 1024, GP = GT_BYTES / 1024, XPL = XP / NLW, GPL = GP / NLW, PPL = XPL + GPL;      // pieces per loader and stage
    constexpr int SPR = GROW / 16, RPP = 1024 / GROW;                   // G strip: 16-byte slots per row, rows per 1 KiB piece
    static_assert(XP % NLW == 0 && GP % NLW == 0 && GPL >= 1 && GPL <= 4, "loader split");
    typedef __attribute__((ext_vector_type(4))) short v4s;
    // G strip in LDS: 16-byte slot s of row r sits at slot s ^ gswz(r), so that the four rows a transposing read touches per quad fall
    // into four different 64-byte bank windows (row pitch 512 / 256 B: rows are bank-aligned, the two low row bits pick the window;
    // pitch 128 B -- the narrow strip -- : odd rows are half a bank cycle further by themselves, row bit 1 picks the other half)
    auto gswz = [](int r) { return SPR >= 16 ? ((r & 3) << 2) : (((r >> 1) & 1) << 2); };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbeg = bz * a.rows_per_split;
    const int rend = min(a.M, rbeg + a.rows_per_split);
    const int nstage = (rend - rbeg + WG_SR - 1) / WG_SR;
    const int gcol0 = jt0 * 16;

    if (SC && wave >= NCW) {
        // ------------------------------------------------------------------ loader waves, row-weighted variant (round 3)
        // The G strip (the stored s of the output layer) does NOT go through LDS-DMA here: a loader fetches its pieces into REGISTERS
        // (plain 16-byte loads, three stages ahead, four register sets), multiplies them by the row weight on the way -- one weight per
        // lane and piece: a lane's 16 bytes are 8 features of ONE data row -- and stores them into the stage's buffer (ds_write_b128),
        // one stage ahead of the compute waves.  Round 2 DMA'd the strip and then scaled it IN PLACE in LDS (ds_read, 20 vector
        // instructions, ds_write, behind a wait for the DMA it had just issued around): that read-modify-write sat on the loaders'
        // serial path, which is what a stage took (60 us alone against 31 for the unweighted kernel on the same shape).  Now the
        // multiplies ride in the issue slots the compute waves' MFMAs leave free, nothing is read back from LDS, and a loader waits for
        // nothing younger than two stages.  Same arithmetic as before: dl = bf16(g_r * s), one rounding.
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        // The loaders' stream (8 memory instructions, ~90 vector instructions of weighting, 4 LDS stores per stage) is what a stage waits
        // for, and as the youngest waves of their SIMDs they lose every issue arbitration against the two compute waves beside them
        // (phase stamps: 870 cycles for the weighting): issued first, they are out of the way instead.
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - NCW;
        const char* zsrc = a.zero + (lane & 31) * 16;
        const int xstride = WG_SR * a.ldX * 2, gstride = WG_DBG(a, 32) ? 0 : WG_SR * a.ldG * 2;
        const char* xbase[XPL]; int xlim[XPL];
#pragma unroll
        for (int i = 0; i < XPL; ++i) {
            const int pc = lw * XPL + i, rl = 2 * pc + (lane >> 5);
            const int col = ((lane & 31) ^ ((rl & 3) << 2)) * 8;                 // source 16-byte chunk of this LDS slot
            xbase[i] = (const char*)a.X + ((size_t)(rbeg + rl) * a.ldX + col) * 2;
            xlim[i] = col < a.ldX ? rend - rbeg - rl : -(1 << 30);
        }
        const char* gbase[GPL]; int glim[GPL];
#pragma unroll
        for (int i = 0; i < GPL; ++i) {
            const int pc = lw * GPL + i, rl = RPP * pc + lane / SPR;
            const int col = gcol0 + ((lane % SPR) ^ gswz(rl)) * 8;
            const bool ok = col < a.ldG;                                          // (a last column block narrower than its strip)
            gbase[i] = (const char*)a.G + ((size_t)(rbeg + rl) * a.ldG + (ok ? col : 0)) * 2;
            glim[i] = ok ? rend - rbeg - rl : -(1 << 30);
        }
        // What a stage costs a CU is the NUMBER of vector-memory wave-instructions, not their bytes (phase stamps of this kernel: 12
        // instructions per loader and stage took 1 430 cycles, ~30 per instruction and CU whatever the width -- the "60-70 GB/s per CU"
        // of DESIGN item 5 is 1 KiB per ~31 cycles).  So the row weights do NOT come by vector loads (4 dword loads per loader and stage
        // = a third of all its instructions for 32 bytes): a loader's G pieces cover NR consecutive data rows, whose weights are ONE
        // scalar load (s_load_dwordx8 through the scalar cache, no vector-memory slot), requested at the top of the iteration that
        // uses them and waited for behind the stage's vector-memory wait.
        constexpr int NR = RPP * GPL;           // data rows of a stage this loader's G pieces cover: rows NR*lw .. NR*lw + NR - 1
        static_assert(NR == 8 || NR == 4, "row weights of a loader and stage are one s_load_dwordx8 / x4");
        typedef __attribute__((ext_vector_type(NR))) float wvec_t;
        auto load_w = [&](int st, wvec_t& w) {
            const uint64_t ad = (uint64_t)(a.rowscale + (rbeg + st * WG_SR + NR * lw));      // (< Mp: the pad rows of gx are finite)
            const uint64_t au = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ad >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ad);
            if constexpr (NR == 8) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(w) : "s"(au) : "memory");
            else asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(w) : "s"(au) : "memory");
        };
        auto wsel = [&](const wvec_t& w, int i) {       // the weight of the lane's row in piece i: row RPP * i + lane / SPR of the loader's NR
            float f = w[RPP * i];
#pragma unroll
            for (int r = 1; r < RPP; ++r) f = (lane / SPR == r) ? w[RPP * i + r] : f;
            return f;
        };
        constexpr int PER = GPL + XPL;          // vector-memory operations per loader and stage: GPL strip pieces, XPL DMA pieces
        u32x4 gq0[GPL], gq1[GPL], gq2[GPL], gq3[GPL];
        // rows beyond the split (and the columns beyond the layer) read a valid address and are multiplied by zero in put_piece
        auto issue_g = [&](int st, int i, u32x4& dst) {
            const char* pg = (st * WG_SR < glim[i]) ? gbase[i] + (size_t)st * (size_t)gstride : (const char*)a.G;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(pg) : "memory");
        };
        auto issue_x = [&](int st, int i) {
            if (WG_DBG(a, 1)) return;
            const char* src = (st * WG_SR < xlim[i]) ? xbase[i] + (size_t)st * (size_t)xstride : zsrc;
            glds16(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + (st % WG_NST) * BUF) + (uint32_t)(lw * XPL + i) * 1024u)));
        };
        auto put_piece = [&](int st, int i, const u32x4& v, const wvec_t& w) {
            const float f = (st * WG_SR < glim[i]) ? (WG_DBG(a, 8) ? 1.0f : wsel(w, i)) : 0.0f;
            *(uint4*)(smem + (st % WG_NST) * BUF + (XP + lw * GPL + i) * 1024 + lane * 16) =
                make_uint4(pack2(bflo(v.x) * f, bfhi(v.x) * f), pack2(bflo(v.y) * f, bfhi(v.y) * f),
                           pack2(bflo(v.z) * f, bfhi(v.z) * f), pack2(bflo(v.w) * f, bfhi(v.w) * f));
        };
        constexpr int XDMA = XPL;
        auto per_stage = [&](int st) { return (st < nstage) ? (WG_DBG(a, 1) ? PER - XDMA : PER) : 0; };
        // prologue: stages 0, 1, 2 requested; stage 0 in its buffer behind the first barrier.  Stage st lives in register set st & 3.
        wvec_t wcur;
        load_w(0, wcur);
        auto issue_all = [&](int st, u32x4 (&gs)[GPL]) {
#pragma unroll
            for (int i = 0; i < GPL; ++i) issue_g(st, i, gs[i]);
#pragma unroll
            for (int i = 0; i < XPL; ++i) issue_x(st, i);
        };
        if (0 < nstage) issue_all(0, gq0);
        if (1 < nstage) issue_all(1, gq1);
        if (2 < nstage) issue_all(2, gq2);
        wait_vmem_but_ws(per_stage(1) + per_stage(2));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wcur) : : "memory");
        if (0 < nstage) {
#pragma unroll
            for (int i = 0; i < GPL; ++i) { asm volatile("" : "+v"(gq0[i])); put_piece(0, i, gq0[i], wcur); }
        }
        __syncthreads();
        WS_STAMP(0)      // prologue
        // Iteration c: the buffer stage c - 1 has left is requested again (stage c + 3), then stage c + 1 (requested two iterations ago) is
        // weighted and stored into its buffer.  (Taking the two in turns, one memory instruction then one piece's weighting, was measured
        // and is slower: 53 us alone against 51, the loaders' iteration 3 080 cycles against 2 030 in the stamped build.)
        auto iter = [&](int c, u32x4 (&gn)[GPL], u32x4 (&gc)[GPL]) {      // n: register set of stage c + 3, c: of stage c + 1
            const bool next = c + 1 < nstage;
            if (next) load_w(c + 1, wcur);
            if (c + 3 < nstage) issue_all(c + 3, gn);
            WS_STAMP(1)      // requests of stage c + 3
            if (next) {
                wait_vmem_but_ws(per_stage(c + 2) + per_stage(c + 3));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wcur) : : "memory");
            }
            WS_STAMP(2)      // wait for stage c + 1
            if (next) {
#pragma unroll
                for (int i = 0; i < GPL; ++i) { asm volatile("" : "+v"(gc[i])); put_piece(c + 1, i, gc[i], wcur); }
            }
            WS_STAMP(3)      // weighting + LDS stores of stage c + 1
            __syncthreads();
            WS_STAMP(4)      // barrier
        };
        for (int c = 0; c < nstage; c += 4) {
            iter(c, gq3, gq1);
            if (c + 1 < nstage) iter(c + 1, gq0, gq2);
            if (c + 2 < nstage) iter(c + 2, gq1, gq3);
            if (c + 3 < nstage) iter(c + 3, gq2, gq0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_STAMP_OUT()
        return;
    }
    if (wave >= NCW) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - NCW;
        const char* zsrc = a.zero + (lane & 31) * 16;
        const char* pbase[PPL];
        int plim[PPL];
        const int xstride = WG_SR * a.ldX * 2, gstride = WG_SR * a.ldG * 2;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const bool isx = i < XPL;
            int rl, col, ld;
            if (isx) {
                const int pc = lw * XPL + i;
                rl = 2 * pc + (lane >> 5);
                col = ((lane & 31) ^ ((rl & 3) << 2)) * 8;               // source 16-byte chunk of this LDS slot
                ld = a.ldX;
            } else {
                const int pc = lw * GPL + (i - XPL);
                rl = RPP * pc + lane / SPR;
                col = gcol0 + ((lane % SPR) ^ gswz(rl)) * 8;
                ld = a.ldG;
            }
            pbase[i] = (const char*)(isx ? a.X : a.G) + ((size_t)(rbeg + rl) * ld + col) * 2;
            plim[i] = col < ld ? rend - rbeg - rl : -(1 << 30);
        }
        auto issue = [&](int st) {
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const bool isx = i < XPL;
                const int pc = isx ? lw * XPL + i : XP + lw * GPL + (i - XPL);
                const char* src = (st * WG_SR < plim[i]) ? pbase[i] + (size_t)st * (size_t)(isx ? xstride : gstride) : zsrc;
                glds16(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr_of(smem + (st % WG_NST) * BUF) + (uint32_t)pc * 1024u)));
            }
        };
        // prologue: P(0) P(1) P(2); stage 0 landed behind the first barrier
        if (0 < nstage) issue(0);
        if (1 < nstage) issue(1);
        if (2 < nstage) issue(2);
        wait_vmem_but_ws((1 < nstage ? PPL : 0) + (2 < nstage ? PPL : 0));
        __syncthreads();
        // iteration c: the buffer stage c - 1 has left is refilled with stage c + 3; the wait for stage c + 1 leaves P(c+2), P(c+3) in flight
        for (int c = 0; c < nstage; ++c) {
            const bool refill = c + 3 < nstage && !WG_DBG(a, 1);
            if (refill) issue(c + 3);
            if (c + 1 < nstage) wait_vmem_but_ws((c + 2 < nstage ? PPL : 0) + (refill ? PPL : 0));
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int q = lane >> 4, l16 = lane & 15, qp = l16 >> 2, p = l16 & 3;
    const int ig = wave % IGC, jg = wave / IGC;        // the wave's i-tiles ig*7 .. +6, local j-tiles jg*BJ .. +BJ-1
    f32x4 acc[AI][BJ], accb[BJ];
#pragma unroll
    for (int u = 0; u < BJ; ++u) {
        accb[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < AI; ++t) acc[t][u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    const int xrow_off = (4 * q + qp) * 512, grow_off = (4 * q + qp) * GROW;
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);      // bf16 1.0 x 8
    const bool do_bias = ig == 0 && !WG_DBG(a, 4);
    __syncthreads();
    WS_STAMP(0)          // prologue
    for (int c = 0; c < nstage; ++c) {
        const int buf = c % WG_NST;
        const char* xb = smem + buf * BUF + xrow_off + p * 16;
        const char* gbase = smem + buf * BUF + XT_BYTES + grow_off;
        uint4 g[BJ];
#pragma unroll
        for (int u = 0; u < BJ; ++u) {
            const int jl = jg * BJ + u;
            const char* gb = gbase + ((((4 * (jl >> 1)) ^ gswz(qp)) + p) * 16) + 8 * (jl & 1);
            const v4s g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)gb);
            const v4s g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(gb + 16 * GROW));
            const uint2 glo = __builtin_bit_cast(uint2, g0), ghi = __builtin_bit_cast(uint2, g1);
            g[u] = make_uint4(glo.x, glo.y, ghi.x, ghi.y);
        }
        WS_STAMP(1)      // G fragments read
        if (!WG_DBG(a, 2)) {
            lds_pipeline<AI, 3>(
                [&](int t) {       // A fragment of i-tile i: P chunk 4*(i>>1)+p (swizzled by the row), half i&1
                    const int i = ig * AI + t;
                    const char* p0 = xb + ((((4 * (i >> 1)) ^ (qp << 2))) * 16) + 8 * (i & 1);
                    const v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
                    const v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p0 + 16 * 512));
                    const uint2 lo = __builtin_bit_cast(uint2, r0), hi = __builtin_bit_cast(uint2, r1);
                    return make_uint4(lo.x, lo.y, hi.x, hi.y);
                },
                [&](int t, const uint4& av) {
#pragma unroll
                    for (int u = 0; u < BJ; ++u) acc[t][u] = mfma16(av, g[u], acc[t][u]);
                });
        }
        if (do_bias) {      // column sums of G through the matrix pipe: every row of ones x G is sum_r G[r][j] (wave-uniform branch)
#pragma unroll
            for (int u = 0; u < BJ; ++u) accb[u] = mfma16(ones, g[u], accb[u]);
        }
        WS_STAMP(2)      // A fragments + MFMAs issued
        __syncthreads();
        WS_STAMP(3)      // barrier
    }
    if (WG_DBG(a, 16)) return;
    // D: lane(col j = lane&15, quad q) reg ii -> out[i = 16*it + 4q + ii][j]
    float* slab = a.slabW + (size_t)bz * a.IT * 16 * a.JT * 16;
#pragma unroll
    for (int u = 0; u < BJ; ++u) {
        const int jt = jt0 + jg * BJ + u;
        if (jt < a.JT) {
#pragma unroll
            for (int t = 0; t < AI; ++t) {
                const int it = ig * AI + t;
                if (it < a.IT) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        slab[(size_t)(it * 16 + 4 * q + ii) * (a.JT * 16) + jt * 16 + l16] = acc[t][u][ii];
                }
            }
            if (ig == 0 && q == 0) a.slabB[(size_t)bz * a.JT * 16 + jt * 16 + l16] = accb[u][0];
        }
    }
    WS_STAMP(4)          // slab stores issued
    WS_STAMP_OUT()
}

template <bool SC, int BJ, int NLW>
__global__ __launch_bounds__((8 + NLW) * 64, NLW == 8 ? 4 : 3) void wgradws_kernel(WgradPArgs a) {
    int bx = blockIdx.x, bz = blockIdx.z;
    wgradws_block(bx, bz, gridDim.x, gridDim.z);
    const int sid = bz * (int)gridDim.x + bx;
    if constexpr (BJ == 4) {
        // a last column block of <= 64 real out-features (the reference's 784 pixels = 3 x 256 + 64, padded to 832): its workgroups
        // take the 64-wide strip shape -- 4 instead of 16 G pieces per stage, 7 instead of 28 MFMAs per wave -- instead of fetching and
        // multiplying 192 columns of zeros (workgroup-uniform branch)
        // (round 4: the unweighted kernel too, where it is a remainder block of a wider layer: the pre-weighted output layer)
        if (bx == (int)gridDim.x - 1 && a.JT - bx * 16 <= 4 && (SC || gridDim.x > 1)) { wgradws_body<SC, 1, NLW>(a, bx * 16, bz, sid); return; }
    }
    wgradws_body<SC, BJ, NLW>(a, bx * 4 * BJ, bz, sid);
}
// two (or three) of them in one launch: blockIdx.z runs over the concatenated row splits
__global__ __launch_bounds__(768, 3) void wgradws_group_kernel(WgradPGroup g) {
    int l = 0;
    while (l + 1 < g.n && (int)blockIdx.z >= g.zbeg[l + 1]) ++l;
    if ((int)blockIdx.x >= g.gx[l]) return;          // (uniform per workgroup: no barrier is skipped by part of a workgroup)
    int bx = blockIdx.x, bz = blockIdx.z - g.zbeg[l];
    wgradws_block(bx, bz, g.gx[l], g.zbeg[l + 1] - g.zbeg[l]);
    wgradws_body<false, 4, 4>(g.a[l], bx * 16, bz, 0);
}

// Several small weight gradients in ONE launch (the three layers of an encoder block over B rows are ~30-130 blocks
// each and latency-bound): blockIdx.z runs over the concatenated row splits of the layers.
__global__ __launch_bounds__(512, 2) void wgradp_group_kernel(WgradPGroup g) {
    int l = 0;
    while (l + 1 < g.n && (int)blockIdx.z >= g.zbeg[l + 1]) ++l;
    if ((int)blockIdx.x >= g.gx[l] || (int)blockIdx.y >= g.gy[l]) return;
    wgradp_body<8, 4, 4, 4, false>(g.a[l], blockIdx.x, blockIdx.y, blockIdx.z - g.zbeg[l]);
}
// ... with row weights on member 0 (the decoder's three layers at small row counts: the output layer's gradient is g2^T (g_r s))
__global__ __launch_bounds__(512, 2) void wgradp_group_sc_kernel(WgradPGroup g) {
    int l = 0;
    while (l + 1 < g.n && (int)blockIdx.z >= g.zbeg[l + 1]) ++l;
    if ((int)blockIdx.x >= g.gx[l] || (int)blockIdx.y >= g.gy[l]) return;
    if (l == 0) wgradp_body<8, 4, 4, 4, true>(g.a[0], blockIdx.x, blockIdx.y, blockIdx.z);
    else wgradp_body<8, 4, 4, 4, false>(g.a[l], blockIdx.x, blockIdx.y, blockIdx.z - g.zbeg[l]);
}

// ---------------------------------------------------------------------------------
// elementwise / reduction kernels
// ---------------------------------------------------------------------------------
// x fp32 [B][X] -> bf16 P-layout [B][Xp] (pads written as zero).
// block = 64 rows (lanes) x 4 chunk-waves; a thread converts one 8-feature P chunk of one row.
// cond != null: C more features per row taken from cond [B][C] (conditional model: concat(x, y), tasks/task05.py:113).
__global__ __launch_bounds__(256) void prep_rows_kernel(const float* x, const float* cond, int B, int X, int C, int Xp, int Bp, uint16_t* XP) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 64 + lane;
    const int nchunk = Xp / 8;
    if (b >= Bp) return;
    for (int c = blockIdx.y * 4 + (threadIdx.x >> 6); c < nchunk; c += gridDim.y * 4) {
        const int t = c >> 2, qq = c & 3;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int f0 = 32 * t + 16 * h + 4 * qq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = f0 + i;
                float t = 0.0f;
                if (b < B && f < X) t = x[(size_t)b * X + f];
                else if (b < B && f < X + C) t = cond[(size_t)b * C + (f - X)];
                v[4 * h + i] = t;
            }
        }
        if (b < B) *(uint4*)(XP + (size_t)b * Xp + 8 * c) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}

// Fused batch gather + dynamic binarisation (main.py:117-120 + src/utils.py:26-27 done per batch on the
// device): row b of the batch is image idx[start+b] of the uint8 dataset resident in HBM; pixel f is
// 1 iff (philox(seed, epoch, image, f/4)[f%4] >> 8) < T(gray), T(g) = floor(g * 2^24 / 255 + 0.5), i.e.
// Bernoulli(gray/255) with an integer threshold (bit-exactly reproducible on the host).  Keyed by
// (epoch, image): one binarisation per image per epoch, as in the reference.  Same block shape and
// output as prep_rows_kernel (P-layout) + optional float32 copy.
// labels != null (conditional models, tasks/task05.py:296-322: the training set is (x, y) pairs): the image's class y rides along -- features
// X .. X + C - 1 of the row are onehot(y) (the encoder's input concat(x, onehot(y)), task05.py:113), and cond_out [B][C] gets the same
// one-hot row as float32 (what iwae_set_condition would have been handed for a host-fed batch).
__global__ __launch_bounds__(256) void gather_binarize_kernel(const uint8_t* data, const int32_t* order, int start, int N, int B, int X, int Xp,
                                                              int Bp, uint64_t seed, uint32_t epoch, uint16_t* XP, float* xf,
                                                              const uint8_t* labels, int C, float* cond_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 64 + lane;
    const int nchunk = Xp / 8;
    if (b >= Bp) return;
    const int img = (b < B) ? order[start + b] : 0;
    const int lab = (labels && b < B) ? (int)labels[img] : -1;
    if (cond_out && blockIdx.y == 0 && threadIdx.x < 64 && b < B)
        for (int j = 0; j < C; ++j) cond_out[(size_t)b * C + j] = (j == lab) ? 1.0f : 0.0f;
    for (int c = blockIdx.y * 4 + (threadIdx.x >> 6); c < nchunk; c += gridDim.y * 4) {
        const int t = c >> 2, qq = c & 3;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int f0 = 32 * t + 16 * h + 4 * qq;
            uint32_t r[4];
            philox4x32_10((uint32_t)img, 0x42494E41u /* 'BINA' */, (uint32_t)(f0 >> 2), epoch, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float bit = 0.0f;
                if (b < B && f0 + i < X) {
                    const uint32_t g = data[(size_t)img * X + f0 + i];
                    const uint32_t thr = (uint32_t)(((uint64_t)g * 16777216ull * 2ull + 255ull) / 510ull);   // floor(g*2^24/255 + 0.5)
                    bit = ((r[i] >> 8) < thr) ? 1.0f : 0.0f;
                    if (xf) xf[(size_t)b * X + f0 + i] = bit;
                } else if (lab >= 0 && f0 + i == X + lab) bit = 1.0f;
                v[4 * h + i] = bit;
            }
        }
        if (b < B) *(uint4*)(XP + (size_t)b * Xp + 8 * c) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}

// The N(0,1) draws of one latent layer for a whole step, fp32 [rows][ld] (4 per thread).  They depend on nothing but
// the counters, so the host launches this a whole step ahead, on the side stream beside the forward pass: the ~40
// quarter-rate integer multiplies per Philox call are off every dependency chain.
// Grid-stride: drawn ahead (a whole step early, beside the forward pass) the launch is a few hundred small blocks that
// take their time in a corner of every CU; as 5 000 blocks it filled the machine for 11 us and the forward's large
// workgroups queued behind it.
// The draws of `nsteps` consecutive steps in one launch (few rows: a step's 2 000 draws are a 4.5 us launch on a chain of 9-17 us kernels; eight steps'
// worth cost the same): step e.step + s goes to out + s * step_stride, each value exactly what eps_gen_kernel would draw for that step.
__global__ __launch_bounds__(256) void eps_gen_multi_kernel(EpsSrc e, int M, int nd4, int ld, float* out, int nsteps, size_t step_stride) {
    const size_t per = (size_t)M * nd4, total = per * nsteps;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int s = (int)(idx / per);
        const size_t r = idx - (size_t)s * per;
        const int row = (int)(r / nd4), d4 = (int)(r - (size_t)row * nd4);
        float n[4];
        normal4(e.row_offset + (uint64_t)row, (uint32_t)d4, e.stream, e.step + (uint32_t)s, e.seed, n);
        *(float4*)(out + (size_t)s * step_stride + (size_t)row * ld + 4 * d4) = make_float4(n[0], n[1], n[2], n[3]);
    }
}
__global__ __launch_bounds__(256) void eps_gen_kernel(EpsSrc e, int M, int nd4, int ld, float* out) {
    const size_t total = (size_t)M * nd4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / nd4), d4 = (int)(idx - (size_t)row * nd4);
        float n[4];
        uint64_t grow = e.row_offset + (uint64_t)row;
        if (e.k_total) { const int b = row / e.kc; grow = e.row_offset + (uint64_t)b * (uint64_t)e.k_total + (uint64_t)(e.s_off + row - b * e.kc); }
        normal4(grow, (uint32_t)d4, e.stream, e.step, e.seed, n);
        *(float4*)(out + (size_t)row * ld + 4 * d4) = make_float4(n[0], n[1], n[2], n[3]);
    }
}

// z = mu + sigma*eps, prior/posterior log-densities, z written bf16 P-layout.
// wave = 16 data rows x 4 quads.  In every 32-feature step t lane (r, q) owns P-layout chunk 4t+q, i.e. features
// 32t+4q..+3 and 32t+16+4q..+3: float4 loads of eps / mu / sigma, one 16-byte store per step, and the row sums
// meet across the 4 quads with two shuffles -- no LDS, no barrier, 800 small blocks instead of 800 of 1024 threads.
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * 4 + wave) * 16 + r;
    const bool valid = row < a.M;
    const int rowc = valid ? row : a.M - 1;        // loads come from a clamped row, results are masked
    const int b = rowc / a.k, s = rowc - b * a.k;
    const float* hd = a.head + (size_t)(a.head_per_row ? rowc : b) * a.ldH;
    float lp = 0.0f, lq = 0.0f, lq2 = 0.0f;
    const int nt = a.Dp / 32;
    for (int t = 0; t < nt; ++t) {
        float z8[8];
        sample_step(a, t, q, rowc, b, s, hd, z8, lp, lq, lq2);
        if (valid && a.ZP)
            *(uint4*)(a.ZP + (size_t)row * a.Dp + 8 * (4 * t + q)) = make_uint4(pack2(z8[0], z8[1]), pack2(z8[2], z8[3]), pack2(z8[4], z8[5]), pack2(z8[6], z8[7]));
        if (valid && a.ZF) {       // float32 mode: z in natural feature order, unrounded
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int f0 = 32 * t + 16 * h + 4 * q;
                // (one 16-byte store per lane where the row pitch allows it: as four dword stores under four conditions the evaluator's sampling
                // kernel took 225 us for 416 k rows -- 64 scattered 4-byte pieces per instruction)
                if ((a.ldZF & 3) == 0 && f0 + 3 < a.ldZF) {
                    *(float4*)(a.ZF + (size_t)row * a.ldZF + f0) = make_float4(z8[4 * h], z8[4 * h + 1], z8[4 * h + 2], z8[4 * h + 3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (f0 + i < a.ldZF) a.ZF[(size_t)row * a.ldZF + f0 + i] = z8[4 * h + i];
                }
            }
        }
    }
    lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
    lq += __shfl_xor(lq, 16); lq += __shfl_xor(lq, 32);
    lq2 += __shfl_xor(lq2, 16); lq2 += __shfl_xor(lq2, 32);
    if (q == 0 && valid) {
        if (a.lp_prior) a.lp_prior[row] = lp;
        a.lq[row] = lq;
        if (a.lq_dreg) a.lq_dreg[row] = lq2;
    }
}

// thread per data row: sum_d log N(z1; mup[row], sigp[row]) with z1 recomputed from the encoder head
// and eps (iwae2.py:122); also the per-row gradients wrt the dec2 head when G != null is done in
// gauss_bwd_kernel.
__global__ __launch_bounds__(256) void gauss_lp_kernel(GaussLpArgs a) {
    // wave = 16 data rows x 4 quads: quad q takes feature groups d4 = q, q+4, ... (float4 loads), two shuffles add them up
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * 4 + wave) * 16 + r;
    const bool valid = row < a.M;
    const int rowc = valid ? row : a.M - 1;
    const int b = rowc / a.k, s = rowc - b * a.k;
    const float* hz = a.zhead + (size_t)b * a.ldZH;           // generating head (per image)
    const float* hp = a.phead + (size_t)rowc * a.ldPH;        // evaluating head (per row)
    float lp = 0.0f;
    const int nd4 = (a.D + 3) / 4;
    for (int d4 = q; d4 < nd4; d4 += 4) {
        float e[4];
        eps4(a.eps, b, s, rowc, d4, a.D, e);
        const float4 zm = *(const float4*)(hz + 4 * d4), zs = *(const float4*)(hz + a.Dzp + 4 * d4);
        const float4 pm = *(const float4*)(hp + 4 * d4), ps = *(const float4*)(hp + a.Dpp + 4 * d4);
        const float zmv[4] = {zm.x, zm.y, zm.z, zm.w}, zsv[4] = {zs.x, zs.y, zs.z, zs.w};
        const float pmv[4] = {pm.x, pm.y, pm.z, pm.w}, psv[4] = {ps.x, ps.y, ps.z, ps.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (4 * d4 + i < a.D) {
                const float z = zmv[i] + zsv[i] * e[i];
                const float u = (z - pmv[i]) * __builtin_amdgcn_rcpf(psv[i]);
                lp += -0.5f * u * u - 0.5f * LOG2PI_F - __logf(psv[i]);
            }
        }
    }
    lp += __shfl_xor(lp, 16);
    lp += __shfl_xor(lp, 32);
    if (q == 0 && valid) a.out[row] = lp;
}

// one wave per image b
__global__ void lse_kernel(LseArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= a.B) return;
    lse_image(a, b, lane, LseGlobalSrc{});
}
// one workgroup of NW waves per image (many samples per image: the evaluator's k)
template <int NW>
__global__ __launch_bounds__(64 * NW) void lse_block_kernel(LseArgs a) {
    __shared__ float slots[NW];
    lse_image(a, (int)blockIdx.x, (int)threadIdx.x, LseGlobalSrc{}, BlockRed<NW>{slots});
}

// Batch means of the per-image values (the scalar entries of the reference's result dict), one 256-thread block.
// A cross-XCD "last wave reduces" inside lse_kernel would need agent-scope release/acquire fences, i.e. an L2
// write-back per wave (measured: 7 -> 71 us); a kernel boundary does that once, so the means are taken either by
// this tiny kernel (forward-only calls) or by an extra block of the step's last kernel (reduce_grads_kernel).
__device__ __forceinline__ void batch_means_block(const float* per_b, int B, float beta, float* out) {
    __shared__ float red[PB_COUNT][4];
    float acc[PB_COUNT];
#pragma unroll
    for (int t = 0; t < PB_COUNT; ++t) acc[t] = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256)
#pragma unroll
        for (int t = 0; t < PB_COUNT; ++t) acc[t] += per_b[t * B + b];
#pragma unroll
    for (int t = 0; t < PB_COUNT; ++t) {
        float v = acc[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[t][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s[PB_COUNT];
        for (int t = 0; t < PB_COUNT; ++t) s[t] = (red[t][0] + red[t][1] + red[t][2] + red[t][3]) / (float)B;
        out[SC_IWAE_ELBO] = s[PB_LME];
        out[SC_VAE_ELBO] = s[PB_MEAN];
        out[SC_IWAE_EQ14] = s[PB_EQ14];
        out[SC_VAE_ELBO_KL] = s[PB_PX] - beta * s[PB_KL];      // iwae1.py:121
        out[SC_MEAN_LPXZ] = s[PB_PX];
        out[SC_MEAN_T1] = s[PB_T1];
        out[SC_MEAN_T2] = s[PB_T2];
        out[SC_INFERENCE_LOSS] = -s[PB_DREG];                  // tasks/task02.py:76
        out[SC_KL] = s[PB_KL];
    }
}
__global__ __launch_bounds__(256) void scalars_kernel(const float* per_b, int B, float beta, float* out) { batch_means_block(per_b, B, beta, out); }

// single block: means over the batch -> scalars[16]

// one block per image b: reduce the sample axis.  Thread (f4, sg) handles 4 latent
// features and the samples s = sg, sg+SG, ...; partial sums meet in LDS.
//   dz_tot = ca*dz_dec + cz*z + cq*(z-mu)/(sigma+1e-6)^2
//   dmu = sum_s dz_tot + kmu*mu ; dsigma = sum_s (dz_tot*eps + cs/sigma) + ksig*(sigma - 1/sigma)
//   da = dsigma * exp(a) = dsigma * (sigma - 1e-6)
__device__ __forceinline__ float sigp_of(const LatentBwdArgs& a, int b, int f) { return a.prior_head[(size_t)b * a.ldH + a.Dp + f]; }
// FAST: the 1-layer training step on the device's own noise (dz as bf16, draws from the step's cache, N(0,1) prior) -- with the
// sources known at compile time the sample loop's loads are straight-line code and all of them are in flight at once; the general
// body's run-time source switches made the compiler wait for each iteration's loads in turn (22 us alone for 42 MB).
// FAST = 2: the 2-layer model's step (three float32 dz terms, summed here).
template <int FAST>
__global__ __launch_bounds__(256) void latent_bwd_kernel(LatentBwdArgs a) {
    __shared__ float red[256][8];
    const int nf4 = a.Dp / 4;              // <= 32
    const int SG = 256 / nf4;
    const int f4 = threadIdx.x % nf4, sg = threadIdx.x / nf4;
    const int b = blockIdx.x;
    const int f0 = 4 * f4;
    float dmu[4] = {0, 0, 0, 0}, dsg[4] = {0, 0, 0, 0}, dpm[4] = {0, 0, 0, 0}, dps[4] = {0, 0, 0, 0};
    float mu[4] = {0, 0, 0, 0}, sgm[4] = {1, 1, 1, 1};
    const bool act = b < a.B && f0 < a.D && sg < SG;
    if (act) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = f0 + i < a.D;
            mu[i] = ok ? a.head[(size_t)b * a.ldH + f0 + i] : 0.0f;
            sgm[i] = ok ? a.head[(size_t)b * a.ldH + a.Dp + f0 + i] : 1.0f;
        }
        float rs2[4], rsg[4];
        float pmu[4] = {0, 0, 0, 0}, prs[4] = {1, 1, 1, 1};       // prior mean and 1/sigma_p (N(0,1) unless a head is given)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s2 = sgm[i] + 1e-6f;
            rs2[i] = 1.0f / (s2 * s2);
            rsg[i] = 1.0f / sgm[i];
            if (!FAST && a.prior_head && f0 + i < a.D) {
                pmu[i] = a.prior_head[(size_t)b * a.ldH + f0 + i];
                prs[i] = 1.0f / a.prior_head[(size_t)b * a.ldH + a.Dp + f0 + i];
            }
        }
        // the sample loop is latency-bound (3 dependent-free loads, little math): keep UN iterations' loads in flight
        constexpr int UN = (FAST == 2) ? 4 : 8;      // (three float32 terms per sample: 4 samples' loads fill the registers)
        for (int s0 = sg; s0 < a.k; s0 += UN * SG) {
            float4 dz[UN], cf[UN];
            float e[UN][4];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int s = s0 + u * SG;
                const bool ok = s < a.k;
                const int sc = ok ? s : a.k - 1;          // clamped, weighted by 0 below
                const int row = b * a.k + sc;
                if constexpr (FAST == 2) {
                    const float4 t1 = *(const float4*)(a.dz + (size_t)row * a.ldDZ + f0), t2 = *(const float4*)(a.dz2 + (size_t)row * a.ldDZ + f0),
                                 t3 = *(const float4*)(a.dz3 + (size_t)row * a.ldDZ + f0);
                    dz[u] = make_float4(t1.x + t2.x + t3.x, t1.y + t2.y + t3.y, t1.z + t2.z + t3.z, t1.w + t2.w + t3.w);
                    cf[u] = a.cf[row];
                    const float4 ev = *(const float4*)(a.eps.cache + (size_t)row * a.eps.ldC + f0);
                    e[u][0] = ev.x; e[u][1] = ev.y; e[u][2] = ev.z; e[u][3] = ev.w;
                } else if constexpr (FAST == 1) {
                    const uint2 h2 = *(const uint2*)(a.dzh + (size_t)row * a.ldDZ + f0);
                    dz[u] = make_float4(bflo(h2.x), bfhi(h2.x), bflo(h2.y), bfhi(h2.y));
                    cf[u] = a.cf[row];
                    const float4 ev = *(const float4*)(a.eps.cache + (size_t)row * a.eps.ldC + f0);
                    e[u][0] = ev.x; e[u][1] = ev.y; e[u][2] = ev.z; e[u][3] = ev.w;
                } else {
                if (a.dzh) { const uint2 h2 = *(const uint2*)(a.dzh + (size_t)row * a.ldDZ + f0); dz[u] = make_float4(bflo(h2.x), bfhi(h2.x), bflo(h2.y), bfhi(h2.y)); }
                else dz[u] = *(const float4*)(a.dz + (size_t)row * a.ldDZ + f0);
                if (a.dz2) {      // 2-layer model: dz1 = decoder path + direct p(z1|z2) term + path through q(z2|z1)
                    const float4 t2 = *(const float4*)(a.dz2 + (size_t)row * a.ldDZ + f0), t3 = *(const float4*)(a.dz3 + (size_t)row * a.ldDZ + f0);
                    dz[u] = make_float4(dz[u].x + t2.x + t3.x, dz[u].y + t2.y + t3.y, dz[u].z + t2.z + t3.z, dz[u].w + t2.w + t3.w);
                }
                cf[u] = a.cf[row];
                eps4(a.eps, b, sc, row, f4, a.D, e[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                // (a clamped sample weighs 0 -- decided here, behind the loads: zeroing the coefficient registers in the load
                // loop made every iteration wait for its own loads)
                if (s0 + u * SG >= a.k) cf[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                const float dzv[4] = {dz[u].x, dz[u].y, dz[u].z, dz[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (f0 + i < a.D) {
                        const float z = mu[i] + sgm[i] * e[u][i];
                        const float up = (z - pmu[i]) * prs[i];                 // (z - mu_p)/sigma_p; = z for the N(0,1) prior
                        const float t = cf[u].x * dzv[i] + cf[u].y * up * prs[i] + cf[u].z * (z - mu[i]) * rs2[i];
                        dmu[i] += t;
                        dsg[i] += t * e[u][i] + cf[u].w * rsg[i];
                        // cf.y = -dLoss/dlpz: gradient of the prior head (task04.py:124-130), summed over the image's samples
                        dpm[i] -= cf[u].y * up * prs[i];
                        dps[i] -= cf[u].y * (up * up - 1.0f) * prs[i];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = dmu[i]; red[threadIdx.x][4 + i] = dsg[i]; }
    __syncthreads();
    if (sg == 0 && act) {
        for (int g = 1; g < SG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { dmu[i] += red[g * nf4 + f4][i]; dsg[i] += red[g * nf4 + f4][4 + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (f0 + i < a.D) {
                dmu[i] += a.kmu * mu[i];
                dsg[i] += a.ksig * (sgm[i] - 1.0f / sgm[i]);
                dsg[i] *= (sgm[i] - 1e-6f);
            } else { dmu[i] = 0.0f; dsg[i] = 0.0f; }
        }
    }
    if (sg == 0 && b < a.B) {
        if (a.DHP) {
            *(uint2*)(a.DHP + (size_t)b * (2 * a.Dp) + p_pos(f0)) = make_uint2(pack2(dmu[0], dmu[1]), pack2(dmu[2], dmu[3]));
            *(uint2*)(a.DHP + (size_t)b * (2 * a.Dp) + p_pos(a.Dp + f0)) = make_uint2(pack2(dsg[0], dsg[1]), pack2(dsg[2], dsg[3]));
        }
        if (a.DHF) {
            *(float4*)(a.DHF + (size_t)b * (2 * a.Dp) + f0) = make_float4(dmu[0], dmu[1], dmu[2], dmu[3]);
            *(float4*)(a.DHF + (size_t)b * (2 * a.Dp) + a.Dp + f0) = make_float4(dsg[0], dsg[1], dsg[2], dsg[3]);
        }
    }
    if (FAST || !a.prior_head) return;       // block-uniform
    // same reduction for the conditional prior's head: d/dmu_p, d/dsigma_p -> pre-activation of exp (sigma_p - 1e-6)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = dpm[i]; red[threadIdx.x][4 + i] = dps[i]; }
    __syncthreads();
    if (sg != 0 || b >= a.B) return;
    if (act) {
        for (int g = 1; g < SG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { dpm[i] += red[g * nf4 + f4][i]; dps[i] += red[g * nf4 + f4][4 + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (f0 + i < a.D) dps[i] *= (sigp_of(a, b, f0 + i) - 1e-6f);
            else { dpm[i] = 0.0f; dps[i] = 0.0f; }
        }
    }
    if (a.DHP2) {
        *(uint2*)(a.DHP2 + (size_t)b * (2 * a.Dp) + p_pos(f0)) = make_uint2(pack2(dpm[0], dpm[1]), pack2(dpm[2], dpm[3]));
        *(uint2*)(a.DHP2 + (size_t)b * (2 * a.Dp) + p_pos(a.Dp + f0)) = make_uint2(pack2(dps[0], dps[1]), pack2(dps[2], dps[3]));
    }
    if (a.DHF2) {
        *(float4*)(a.DHF2 + (size_t)b * (2 * a.Dp) + f0) = make_float4(dpm[0], dpm[1], dpm[2], dpm[3]);
        *(float4*)(a.DHF2 + (size_t)b * (2 * a.Dp) + a.Dp + f0) = make_float4(dps[0], dps[1], dps[2], dps[3]);
    }
}

// per data row and 4 features, 2-layer model: everything that touches a per-row
// Gaussian head.  mode 0 (dec2 head, after lpz1z2):   u=(z1-mup)/sigp
//     dhead = G*[ u/sigp | (u^2-1)/sigp * exp(ap) ],  dz1_direct = G*(-u/sigp)   -> DZACC (fp32, =)
// mode 1 (enc2 head, after dec2 backward gave dz2_dec):
//     dz2 = dz2_dec - G*z2 ; dhead = [ dz2 | (dz2*eps2 + G/sig2) * exp(a2) ]
__global__ void gauss_bwd_kernel(GaussBwdArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nf4 = a.Dp / 4;
    const int row = idx / nf4, f4 = idx % nf4;
    if (row >= a.Mp) return;
    const bool valid = row < a.M;
    const int f0 = 4 * f4;
    float dm[4] = {0, 0, 0, 0}, ds[4] = {0, 0, 0, 0}, dzd[4] = {0, 0, 0, 0};
    if (valid && f0 < a.D) {
        const int b = row / a.k, s = row - b * a.k;
        const float G = a.G[row];
        const float* hp = a.head + (size_t)row * a.ldH;
        float e[4];
        eps4(a.eps, b, s, row, f4, a.D, e);
        const float4 mu4 = *(const float4*)(hp + f0), sg4 = *(const float4*)(hp + a.Dp + f0);
        const float muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, sgv[4] = {sg4.x, sg4.y, sg4.z, sg4.w};
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;      // mode 0: generating head (per image); mode 1: dz from dec2
        if (a.mode == 0) {
            const float* hz = a.zhead + (size_t)b * a.ldZH;
            a4 = *(const float4*)(hz + f0); b4 = *(const float4*)(hz + a.Dzp + f0);
        } else {
            a4 = *(const float4*)(a.dz_in + (size_t)row * a.ldDZ + f0);
        }
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (f0 + i < a.D) {
                const float mu = muv[i], sg = sgv[i], rs = __builtin_amdgcn_rcpf(sg);
                if (a.mode == 0) {
                    const float z = av[i] + bv[i] * e[i];
                    const float u = (z - mu) * rs;
                    dm[i] = G * u * rs;
                    ds[i] = G * (u * u - 1.0f) * rs * (sg - 1e-6f);
                    dzd[i] = -dm[i];
                } else {
                    const float z = mu + sg * e[i];
                    const float d = av[i] - G * z;
                    dm[i] = d;
                    ds[i] = (d * e[i] + G * rs) * (sg - 1e-6f);
                }
            }
        }
    }
    if (valid) {
        if (a.DHP) {
            *(uint2*)(a.DHP + (size_t)row * (2 * a.Dp) + p_pos(f0)) = make_uint2(pack2(dm[0], dm[1]), pack2(dm[2], dm[3]));
            *(uint2*)(a.DHP + (size_t)row * (2 * a.Dp) + p_pos(a.Dp + f0)) = make_uint2(pack2(ds[0], ds[1]), pack2(ds[2], ds[3]));
        }
        if (a.DHF) {
            *(float4*)(a.DHF + (size_t)row * (2 * a.Dp) + f0) = make_float4(dm[0], dm[1], dm[2], dm[3]);
            *(float4*)(a.DHF + (size_t)row * (2 * a.Dp) + a.Dp + f0) = make_float4(ds[0], ds[1], ds[2], ds[3]);
        }
        if (a.mode == 0) *(float4*)(a.dz_direct + (size_t)row * a.ldDZ + f0) = make_float4(dzd[0], dzd[1], dzd[2], dzd[3]);
    }
}

// out[row][f] = sum of up to 3 fp32 [M][ld] arrays (dz1 contributions of the 2-layer model)

// ---------------------------------------------------------------------------------
// gradient slab reduce and Adam (+ bf16 A-image refresh)
// ---------------------------------------------------------------------------------
struct AdamCoef { float alpha, gscale, beta1, beta2, eps; int on; };

// Keras Adam (epsilon outside the square root, main.py:93) on one element + refresh of the bf16 weight images /
// the fp32 bias block the GEMM kernels read.  is_bias: element j of the bias, else weight (in-feature i, out-feature j).
// The update's arithmetic with its roundings spelled out (explicit fused multiply-adds): the same function is inlined into
// reduce_grads_kernel, adam_kernel and wgrad_rows_kernel, and the single-GPU step must land on bit-identical parameters whichever of them
// applies it (left to -ffp-contract, the compiler fused a different product of b1*m + (1-b1)*g in one of the three contexts).
__device__ __forceinline__ void adam_math(float& w, float& m, float& v, float g, const AdamCoef& c) {
    g *= c.gscale;
    m = __builtin_fmaf(c.beta1, m, (1.0f - c.beta1) * g);
    v = __builtin_fmaf(c.beta2, v, ((1.0f - c.beta2) * g) * g);
    const float q = (c.alpha * m) / (sqrtf(v) + c.eps);
    w = w - q;
}
// refresh of the bf16 weight images / the fp32 bias block the GEMM kernels read, from the (new) master value w
__device__ __forceinline__ void image_refresh(const LayerDesc& L, bool is_bias, int i, int j, float w) {
    if (is_bias) {
        *(float*)(L.imgF + img_mg_bias_byte(L.joff + j, L.KT_F)) = w;
    } else {
        const uint16_t h = (uint16_t)(pack2(w, 0.0f) & 0xffffu);
        *(uint16_t*)(L.imgF + img_mg_byte(L.joff + j, i, L.KT_F)) = h;
        if (L.imgB) {
            if (L.imgB_kmajor) *(uint16_t*)(L.imgB + img_k_byte(i, L.joff + j, L.MT_B)) = h;
            else *(uint16_t*)(L.imgB + img_mg_byte(i, L.joff + j, L.KT_B)) = h;
        }
    }
}
__device__ __forceinline__ void adam_element(const LayerDesc& L, bool is_bias, int i, int j, size_t idx, float g, bool update,
                                             float* param, float* mom, float* vel, const AdamCoef& c) {
    float w = param[idx];
    if (update) {
        float m = mom[idx], v = vel[idx];
        adam_math(w, m, v, g, c);
        mom[idx] = m;
        vel[idx] = v;
        param[idx] = w;
    }
    image_refresh(L, is_bias, i, j, w);
}

// Sums the per-split fp32 slabs of every layer into the flat gradient; with c.on the Adam update of the same
// elements follows in the same thread (single-GPU train step: no all-reduce sits between the two).
struct MeansArgs { const float* per_b; int B; float beta; float* out; };     // per_b == null: no extra block

// The grid covers reduce blocks [first_block, first_block + n) of the layer table: the whole table, or -- single-GPU
// train step -- the decoder's layers on the side stream and the rest on the main stream (see backward_impl).
__global__ __launch_bounds__(256) void reduce_grads_kernel(const LayerDesc* layers, int nlayers, float* grad, float* param, float* mom,
                                                           float* vel, AdamCoef c, MeansArgs mn, int first_block, int n1, int first_block2) {
    if (mn.per_b && blockIdx.x == gridDim.x - 1) {      // the one extra block of the grid: batch means of this step
        batch_means_block(mn.per_b, mn.B, mn.beta, mn.out);
        return;
    }
    // (two block ranges: the layers whose weight gradients ran on one side stream need not be neighbours in the table)
    const int bid = (int)blockIdx.x < n1 ? (int)blockIdx.x + first_block : (int)blockIdx.x - n1 + first_block2;
    // block = 64 groups of 4 consecutive out-features (float4 loads) x 4 split groups; partial sums meet in LDS.
    // A group never straddles a weight row: rblock counts are computed per row of 4-float groups (Nout4 = ceil(Nout/4)).
    __shared__ float4 red[4][64];
    int l = 0;
    while (l + 1 < nlayers && bid >= layers[l + 1].rblock_begin) ++l;
    const LayerDesc L = layers[l];
    const int n4 = (L.Nout + 3) >> 2;                       // float4 groups per weight row
    const int g = (bid - L.rblock_begin) * 64 + (threadIdx.x & 63);
    const int sg = threadIdx.x >> 6;
    const int ngroups = (L.Kin + 1) * n4;                   // Kin weight rows + 1 bias row
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0, j = 0;
    const bool ok = g < ngroups;
    if (ok) {
        i = g / n4;
        j = (g - i * n4) * 4;
        const bool is_b = i == L.Kin;
        const float* p = is_b ? (L.slabB + L.joff + j) : (L.slabW + (size_t)i * L.slab_ld + L.joff + j);
        const size_t stride = is_b ? (L.slabB_stride ? L.slabB_stride : (size_t)L.slab_ld) : L.slab_stride;
        // slab rows start 16-byte aligned (joff, slab_ld multiples of 16 floats): float4 loads; columns >= Nout are pads (zeros / ignored)
        // 8 slabs' loads in flight per thread, summed in slab order (a 64-split layer was 4 round trips of 4 loads: the kernel's time
        // was this chain, not its bytes)
        if (L.nsplit <= 16) {      // few row splits (the encoder's layers): at most 4 loads per thread, all in flight
#pragma unroll 4
            for (int sp = sg; sp < L.nsplit; sp += 4) {
                const float4 v = *(const float4*)(p + (size_t)sp * stride);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        } else
        for (int sp0 = sg; sp0 < L.nsplit; sp0 += 32) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sp = sp0 + 4 * u;
                v[u] = *(const float4*)(p + (size_t)(sp < L.nsplit ? sp : sg) * stride);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (sp0 + 4 * u < L.nsplit) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
    }
    red[sg][threadIdx.x & 63] = s;
    __syncthreads();
    if (sg == 0 && ok) {
        const int t = threadIdx.x;
        float r[4] = {red[0][t].x + red[1][t].x + red[2][t].x + red[3][t].x, red[0][t].y + red[1][t].y + red[2][t].y + red[3][t].y,
                      red[0][t].z + red[1][t].z + red[2][t].z + red[3][t].z, red[0][t].w + red[1][t].w + red[2][t].w + red[3][t].w};
        float* dst = (i == L.Kin) ? (grad + L.offb + j) : (grad + L.offW + (size_t)i * L.Nout + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j + e < L.Nout) {
                dst[e] = r[e];
                if (c.on) adam_element(L, i == L.Kin, i, j + e, (size_t)(dst - grad) + e, r[e], true, param, mom, vel, c);
            }
        }
    }
}

// Keras Adam (main.py:93): theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps), eps = 1e-4;
// then refresh the bf16 A-images the GEMM kernels stream (forward W^T and backward W).
__global__ void adam_kernel(const LayerDesc* layers, int nlayers, float* param, const float* grad, float* mom, float* vel, AdamCoef c, int first_block) {
    const int bid = (int)blockIdx.x + first_block;       // the grid covers blocks [first_block, first_block + gridDim.x) of the layer table
    int l = 0;
    while (l + 1 < nlayers && bid >= layers[l + 1].block_begin) ++l;
    const LayerDesc L = layers[l];
    const int e = (bid - L.block_begin) * blockDim.x + threadIdx.x;
    const int nW = L.Kin * L.Nout;
    if (e >= nW + L.Nout) return;
    const bool is_b = e >= nW;
    const size_t idx = is_b ? (L.offb + (e - nW)) : (L.offW + e);
    adam_element(L, is_b, is_b ? 0 : e / L.Nout, is_b ? e - nW : e % L.Nout, idx, c.on ? grad[idx] : 0.0f, c.on != 0, param, mom, vel, c);
}
// ---------------------------------------------------------------------------------
// wgrad_rows_kernel (round 4): the weight gradients of a BasicBlock applied to FEW rows (the image encoder: R = batch size <= 2 048)
// with the WHOLE row reduction inside one workgroup -- no row splits, no fp32 slabs, no reduction launch -- and the Keras Adam update
// (+ weight-image refresh) of the workgroup's 64 x 64 block of the weight matrix in its epilogue.  Before: wgradp_group_kernel (32 - 48
// workgroups of 256 x 128 features walking row splits: 18.6 us alone) -> slabs -> reduce_grads_kernel (+ Adam: 14.6 us), two dependent
// launches at the END of the step's main chain, in front of the next encoder forward.
//   grid: one workgroup per (linear map, 64 in-features, 64 out-features) [+ one block for the step's batch means]
//   wave w of 4: data rows [w*RW, (w+1)*RW) -- its own ring of 4 stages x 32 rows x (128 B of X + 128 B of G) filled by its own LDS-DMA and
//   read back through ds_read_b64_tr_b16 (the P-layout operands are row-major; the contraction index is the data row): no workgroup
//   barrier in the loop, a wave waits only for its own vmcnt.  16 accumulator tiles (4 i-tiles x 4 j-tiles) per wave + the bias
//   gradient through the matrix pipe (ones x G).  The four waves' partial sums meet in LDS (fixed order: deterministic), then every
//   thread owns 16 elements: gradient -> flat buffer, and (c.on) m, v, theta, the bf16 images.  Reference: tape.gradient + apply_gradients,
//   src/iwae1.py:159-160 (Adam: main.py:93).
// ---------------------------------------------------------------------------------
struct WgradRowsArgs {
    WgradRowsJob job[WGR_MAX_JOBS];
    int njobs;
    const LayerDesc* layers;
    float *grad, *param, *mom, *vel;
    AdamCoef c;
    MeansArgs mn;              // per_b != null: the last block of the grid makes the step's batch means
    const char* zero;          // >= 1 KiB of zeros
};
#define WGR_NST 4
#define WGR_STAGE 8192         // bytes per stage and wave: 32 rows x 128 B of X, 32 rows x 128 B of G
__global__ __launch_bounds__(256, 1) void wgrad_rows_kernel(WgradRowsArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((ext_vector_type(4))) short v4s;
    if (a.mn.per_b && blockIdx.x == gridDim.x - 1) {
        batch_means_block(a.mn.per_b, a.mn.B, a.mn.beta, a.mn.out);
        return;
    }
    int jn = 0;
#pragma unroll
    for (int t = 1; t < WGR_MAX_JOBS; ++t)
        if (t < a.njobs && (int)blockIdx.x >= a.job[t].wg_begin) jn = t;
    const WgradRowsJob J = a.job[jn];
    const int wl = (int)blockIdx.x - J.wg_begin;
    const int bi = wl / J.jb, bj = wl - bi * J.jb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15, qp = l16 >> 2, p = l16 & 3;
    const int RW = ((J.R + 3) / 4 + 31) & ~31;                  // data rows per wave (multiple of the 32-row stage)
    const int rbeg = wave * RW, rend = min(J.R, rbeg + RW);
    const int nstage = rend > rbeg ? (rend - rbeg + 31) / 32 : 0;
    char* ring = smem + wave * (WGR_NST * WGR_STAGE);
    // 16-byte slot s of row r lands at slot s ^ swz(r) (applied on the SOURCE address): at a 128-byte row pitch odd rows are half a bank
    // cycle apart by themselves, row bit 1 picks the other half (the narrow strip of wgradws_kernel)
    auto swz = [](int r) { return ((r >> 1) & 1) << 2; };
    // DMA pieces of a stage: 4 of X (8 rows x 128 B each), 4 of G; a lane fetches the same (row in stage, slot) in every stage
    const int prl = lane >> 3, psl = lane & 7;
    const char* zsrc = a.zero + lane * 16;
    const char* xsrc[4]; const char* gsrc[4]; int xlim[4], glim[4];
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
        const int rl = 8 * pc + prl;
        const int cx = bi * 64 + ((psl ^ swz(rl)) * 8), cg = bj * 64 + ((psl ^ swz(rl)) * 8);
        xsrc[pc] = (const char*)J.X + ((size_t)(rbeg + rl) * J.ldX + cx) * 2;
        gsrc[pc] = (const char*)J.G + ((size_t)(rbeg + rl) * J.ldG + cg) * 2;
        xlim[pc] = cx < J.ldX ? rend - rbeg - rl : -(1 << 30);      // the piece's row is valid in stage c iff 32*c < lim
        glim[pc] = cg < J.ldG ? rend - rbeg - rl : -(1 << 30);
    }
    const size_t xstride = (size_t)32 * J.ldX * 2, gstride = (size_t)32 * J.ldG * 2;
    auto issue = [&](int c) {
        const uint32_t base = lds_addr_of(ring + (c % WGR_NST) * WGR_STAGE);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc)
            glds16(32 * c < xlim[pc] ? xsrc[pc] + (size_t)c * xstride : zsrc, (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + (uint32_t)pc * 1024u)));
#pragma unroll
        for (int pc = 0; pc < 4; ++pc)
            glds16(32 * c < glim[pc] ? gsrc[pc] + (size_t)c * gstride : zsrc, (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + 4096u + (uint32_t)pc * 1024u)));
    };
    f32x4 acc[4][4], accb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        accb[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);      // bf16 1.0 x 8
    const bool do_bias = bi == 0;
    for (int c = 0; c < min(nstage, WGR_NST - 1); ++c) issue(c);
    const int row_off = (4 * q + qp) * 128;
    for (int c = 0; c < nstage; ++c) {
        wait_vmem_but_ws(8 * min(WGR_NST - 2, nstage - 1 - c));     // stage c has landed; the (<= 2) younger stages may still fly
        if (c + WGR_NST - 1 < nstage) issue(c + WGR_NST - 1);        // into the buffer stage c - 1 has left (its reads fed MFMAs already issued)
        const char* xb = ring + (c % WGR_NST) * WGR_STAGE + row_off;
        const char* gb = xb + 4096;
        auto frag = [&](const char* base, int t) {       // tile t of the 64-feature block: P chunk 4*(t>>1)+p (swizzled by the row), half t&1
            const char* p0 = base + ((((4 * (t >> 1)) ^ swz(qp)) + p) * 16) + 8 * (t & 1);
            const v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
            const v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p0 + 16 * 128));
            const uint2 lo = __builtin_bit_cast(uint2, r0), hi = __builtin_bit_cast(uint2, r1);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        };
        uint4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = frag(gb, u);
        if (J.rowscale) {      // row-weighted G (workgroup-uniform): a fragment's 8 data rows are 4q..4q+3 and 16+4q..16+4q+3 of the stage; dl = bf16(g_r * s), one rounding
            const float* rs = J.rowscale + rbeg + 32 * c + 4 * q;
            const float4 s0 = *(const float4*)rs, s1 = *(const float4*)(rs + 16);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                g[u] = make_uint4(pack2(bflo(g[u].x) * s0.x, bfhi(g[u].x) * s0.y), pack2(bflo(g[u].y) * s0.z, bfhi(g[u].y) * s0.w),
                                  pack2(bflo(g[u].z) * s1.x, bfhi(g[u].z) * s1.y), pack2(bflo(g[u].w) * s1.z, bfhi(g[u].w) * s1.w));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint4 av = frag(xb, t);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = mfma16(av, g[u], acc[t][u]);
        }
        if (do_bias) {
#pragma unroll
            for (int u = 0; u < 4; ++u) accb[u] = mfma16(ones, g[u], accb[u]);
        }
    }
    // the waves' partial sums meet in LDS: each wave writes its 16 tiles into its own ring (conflict-free 16-byte stores), the bias row behind them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) *(f32x4*)(ring + ((t * 4 + u) * 64 + lane) * 16) = acc[t][u];
    if (q == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *(float*)(ring + 16384 + (u * 16 + l16) * 4) = accb[u][0];
    }
    __syncthreads();
    // Epilogue.  Thread (wave, lane) owns, of the 64 x 64 block, the i-tiles t = 2s, 2s+1 (s = wave >> 1: one 32-in-feature k-step) and the
    // j-tiles u = 2(wave & 1), +1 with the accumulator's lane map: i = 64 bi + 16t + 4q + ii, j = 64 bj + 16u + l16 -- 16 elements, of which
    // the 8 of one u (two tiles x four in-features) are ONE 16-byte chunk of the forward weight image (layout.h: a lane's 8 values of a
    // k-step).  All sums first, then theta / m / v requested in one batch, the arithmetic, the stores; the new weights go into the images as
    // 16-byte stores -- the forward image straight from registers, the backward image (rows = in-features) through a bf16 tile in LDS.
    // (Element by element with 2-byte image stores the epilogue was 16 dependent round trips and ~100 scattered stores per thread: 15 of
    // the kernel's 18 us.)
    const int jg = bj * 64;
    const LayerDesc L0 = a.layers[J.sub0];
    const LayerDesc L1 = a.layers[J.sub1 >= 0 ? J.sub1 : J.sub0];
    const int s2 = wave >> 1, u0 = 2 * (wave & 1);
    float gsum[16], pw[16], pm[16], pv[16];
    size_t eidx[16];
    bool eok[16];
    char* tileB = smem + 20480;                 // bf16 [64 in-features][64 out-features], pitch 144 B (behind wave 0's tiles and bias row)
    constexpr int TB_PITCH = 144;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
        const int u = u0 + uu;
        const bool second = J.sub1 >= 0 && jg + 16 * u >= J.split;       // (wave-uniform: the split is a multiple of 32 features)
        const int joff = second ? L1.joff : L0.joff, Nout = second ? L1.Nout : L0.Nout, Kin = second ? L1.Kin : L0.Kin;
        const size_t offW = second ? L1.offW : L0.offW;
        const int jj = jg + 16 * u + l16 - joff;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int n = (2 * s2 + tt) * 4 + u;
            f32x4 v = *(const f32x4*)(smem + (n * 64 + lane) * 16);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 o = *(const f32x4*)(smem + w * (WGR_NST * WGR_STAGE) + (n * 64 + lane) * 16);
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = bi * 64 + 32 * s2 + 16 * tt + 4 * q + ii, e = 8 * uu + 4 * tt + ii;
                eok[e] = jj >= 0 && jj < Nout && i < Kin;
                eidx[e] = eok[e] ? offW + (size_t)i * Nout + jj : offW;
                gsum[e] = v[ii];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        pw[e] = a.param[eidx[e]];
        if (a.c.on) { pm[e] = a.mom[eidx[e]]; pv[e] = a.vel[eidx[e]]; }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (eok[e]) {
            a.grad[eidx[e]] = gsum[e];
            if (a.c.on) {
                adam_math(pw[e], pm[e], pv[e], gsum[e], a.c);
                a.mom[eidx[e]] = pm[e];
                a.vel[eidx[e]] = pv[e];
                a.param[eidx[e]] = pw[e];
            }
        } else {
            pw[e] = 0.0f;          // pad rows / columns of the images stay zero
        }
    }
    if (a.c.on) {
        // forward image (rows = out-features): this lane's 8 in-features of k-step (64 bi + 32 s2) / 32 for out-feature j
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int jl = 16 * (u0 + uu) + l16, j = jg + jl;
            const int kf0 = bi * 64 + 32 * s2 + 4 * q;
            const uint4 ch = make_uint4(pack2(pw[8 * uu + 0], pw[8 * uu + 1]), pack2(pw[8 * uu + 2], pw[8 * uu + 3]),
                                        pack2(pw[8 * uu + 4], pw[8 * uu + 5]), pack2(pw[8 * uu + 6], pw[8 * uu + 7]));
            if (j < J.ldG && kf0 < 32 * L0.KT_F) *(uint4*)(L0.imgF + img_mg_byte(j, kf0, L0.KT_F)) = ch;
            if (L0.imgB) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {      // tile [in-feature 32 s2 + 16 tt + 4q + ii][out-feature jl]
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        *(uint16_t*)(tileB + (32 * s2 + 16 * tt + 4 * q + ii) * TB_PITCH + jl * 2) = (uint16_t)(pack2(pw[8 * uu + 4 * tt + ii], 0.0f) & 0xffffu);
                }
            }
        }
        if (L0.imgB) {      // backward image (rows = in-features, k = out-feature space): 512 chunks of 16 bytes per block, two per thread
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = tid + 256 * r, il = c >> 3, sp = (c >> 2) & 1, qq = c & 3;
                const int i = bi * 64 + il, kf0 = jg + 32 * sp + 4 * qq;
                const uint2 lo = *(const uint2*)(tileB + il * TB_PITCH + (32 * sp + 4 * qq) * 2);
                const uint2 hi = *(const uint2*)(tileB + il * TB_PITCH + (32 * sp + 16 + 4 * qq) * 2);
                // (MG-major or K-major: the 16-byte chunk of (row i, k-step quarter) holds the same 8 out-features either way, layout.h)
                if (i < J.ldX && kf0 < J.ldG)
                    *(uint4*)(L0.imgB + (L0.imgB_kmajor ? img_k_byte(i, kf0, L0.MT_B) : img_mg_byte(i, kf0, L0.KT_B))) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
    if (do_bias && tid < 64) {
        float b = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) b += *(const float*)(smem + w * (WGR_NST * WGR_STAGE) + 16384 + tid * 4);
        const int j = jg + tid;
        const bool second = J.sub1 >= 0 && j >= J.split;
        const LayerDesc L = second ? a.layers[J.sub1] : L0;
        const int jj = j - L.joff;
        if (jj >= 0 && jj < L.Nout) {
            const size_t idx = L.offb + jj;
            a.grad[idx] = b;
            if (a.c.on) adam_element(L, true, 0, jj, idx, b, true, a.param, a.mom, a.vel, a.c);
        }
    }
}

// ---------------------------------------------------------------------------------
// exports in the reference's [k, B, ...] order (iwae1.py:141-151), debug dumps
// ---------------------------------------------------------------------------------
__global__ void export_rows_kernel(const float* in, int B, int k, float* out) {   // [B*k] row order -> [k][B]
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B * k) return;
    const int b = r / k, s = r - b * k;
    out[(size_t)s * B + b] = in[r];
}
__global__ void export_z_kernel(SampleArgs a, float* zout, const float* wn, float* snis) {
    // thread per (row, d4): z[k][B][D]; snis_z via atomics-free second kernel below
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nd4 = (a.D + 3) / 4;
    const int row = idx / nd4, d4 = idx % nd4;
    if (row >= a.M) return;
    const int b = row / a.k, s = row - b * a.k;
    const float* hd = a.head + (size_t)(a.head_per_row ? row : b) * a.ldH;
    float e[4];
    eps4(a.eps, b, s, row, d4, a.D, e);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = 4 * d4 + i;
        if (f < a.D) zout[((size_t)s * a.B + b) * a.D + f] = hd[f] + hd[a.Dp + f] * e[i];
    }
}
__global__ void snis_kernel(const float* z, const float* wn, int B, int k, int D, float* out) {   // iwae1.py:139
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx % D;
    float s = 0.0f;
    for (int i = 0; i < k; ++i) s += wn[b * k + i] * z[((size_t)i * B + b) * D + d];
    out[idx] = s;
}
__global__ void unpack_p_kernel(const uint16_t* P, int rows, int F, int Fp, float* out) {   // P-layout bf16 -> [rows][F] fp32
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * F) return;
    const int r = idx / F, f = idx % F;
    out[idx] = __uint_as_float((uint32_t)P[(size_t)r * Fp + p_pos(f)] << 16);
}
__global__ void eps_dump_kernel(EpsSrc e, int B, int k, int D, float* out) {   // [k][B][D]
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nd4 = (D + 3) / 4;
    const int row = idx / nd4, d4 = idx % nd4;
    if (row >= B * k) return;
    const int b = row / k, s = row - b * k;
    float n[4];
    eps4(e, b, s, row, d4, D, n);
    for (int i = 0; i < 4; ++i)
        if (4 * d4 + i < D) out[((size_t)s * B + b) * D + 4 * d4 + i] = n[i];
}

// ---------------------------------------------------------------------------------
// launch wrappers (plain C++ callers in model.cpp)
// ---------------------------------------------------------------------------------
// Completion event riding on a kernel's own dispatch packet (hipExtLaunchKernelGGL stop event) instead of a separate
// hipEventRecord behind it.  Measured on MI355X (tools/probe/event_latency.hip): the record puts a 5.5-7 us bubble in
// front of the recording stream's next kernel and the waiting stream starts 12 us after the kernel ended; with the event
// on the dispatch packet these are 2.2 and 7.6 us, and a join back 6.2 instead of 10.  set_launch_stop_event(e) arms it
// for the NEXT launch that goes through LAUNCH_EV on this thread.
static thread_local hipEvent_t g_stop_event = nullptr;
void set_launch_stop_event(hipEvent_t e) { g_stop_event = e; }
#define LAUNCH_EV(kern, grid, block, lds, st, ...)                                                        \
    do {                                                                                                  \
        hipEvent_t ev_ = g_stop_event;                                                                    \
        g_stop_event = nullptr;                                                                           \
        if (ev_) hipExtLaunchKernelGGL(kern, grid, block, lds, st, nullptr, ev_, 0, __VA_ARGS__);         \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                 \
    } while (0)
static inline dim3 grid1(size_t n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

template <int KTC>
static void launch_dense_k(int epi, const DenseArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    switch (epi) {
        case EPI_TANH: LAUNCH_EV((dense_kernel<EPI_TANH, KTC, 2>), grid, dim3(256), lds, st, a); break;
        case EPI_HEAD: LAUNCH_EV((dense_kernel<EPI_HEAD, KTC, 2>), grid, dim3(256), lds, st, a); break;
        case EPI_DX: LAUNCH_EV((dense_kernel<EPI_DX, KTC, 2>), grid, dim3(256), lds, st, a); break;
        case EPI_F32: LAUNCH_EV((dense_kernel<EPI_F32, KTC, 2>), grid, dim3(256), lds, st, a); break;
        case EPI_BERN: LAUNCH_EV((dense_kernel<EPI_BERN, KTC, 2>), grid, dim3(256), lds, st, a); break;
        case EPI_SIGMOID: LAUNCH_EV((dense_kernel<EPI_SIGMOID, KTC, 2>), grid, dim3(256), lds, st, a); break;
    }
}
// 8 waves x 16 rows (four waves per SIMD): instantiated for the large-row-count launches of the reference shapes
static bool launch_dense_g1(int epi, const DenseArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    // K > 256 (the 784-pixel input layer): the window loop is bound by issuing its LDS-DMA pieces and row loads (phase stamps:
    // 37 % issue + 45 % wait, 3 % MFMA at B = 1024) -- eight waves share that issue work
    if (a.KT > 8 && epi == EPI_TANH) {
        LAUNCH_EV((dense_kernel<EPI_TANH, 0, 1>), grid, dim3(512), a.stage_all ? 4 * DENSE_UNIT : lds, st, a);
        return true;
    }
    if (a.zhead) {      // sampled-input mode (host asks for it only where an instantiation exists: 1-layer training step, latent <= 128)
        if (a.KT == 4) LAUNCH_EV((dense_kernel<EPI_TANH, 4, 1, 8, true>), grid, dim3(512), lds, st, a);
        else LAUNCH_EV((dense_kernel<EPI_TANH, 2, 1, 8, true>), grid, dim3(512), lds, st, a);
        return true;
    }
    if (a.M < 8192 || !((a.g1_mask >> epi) & 1u)) return false;
    // the 200 -> 200 tanh layer (d2): 16 waves x 16 rows, one workgroup per CU -- half the LDS-DMA pieces and row loads per
    // wave again, one weight stream per 256 rows (measured: -5 us per step; the other epilogues and the 100 -> 200 layer: no change)
    if (a.KT == 7 && epi == EPI_TANH) {
        LAUNCH_EV((dense_kernel<EPI_TANH, 7, 1, 16>), dim3((a.M + 255) / 256, grid.y), dim3(1024), lds, st, a);
        return true;
    }
    if (a.KT == 7) {
        switch (epi) {
            case EPI_DX: LAUNCH_EV((dense_kernel<EPI_DX, 7, 1>), grid, dim3(512), lds, st, a); return true;
            case EPI_F32: LAUNCH_EV((dense_kernel<EPI_F32, 7, 1>), grid, dim3(512), lds, st, a); return true;
            case EPI_BERN: LAUNCH_EV((dense_kernel<EPI_BERN, 7, 1>), grid, dim3(512), lds, st, a); return true;
            default: return false;
        }
    }
    if (a.KT == 4 && epi == EPI_TANH) { LAUNCH_EV((dense_kernel<EPI_TANH, 4, 1>), grid, dim3(512), lds, st, a); return true; }
    return false;
}
bool block_fwd_ok(const BlockFwdArgs& a) {
    if (a.sample && (a.S.Dp != 32 * a.KT0 || a.KT0 > 15 || !a.S.ZP || a.S.ZF || a.S.M != a.R)) return false;
    if (a.oimg && (a.NT2 != 0 || a.oH < 1 || !a.oXB || !a.olpxz || a.ok < 1)) return false;
    return a.R <= 4096 && a.KT1 <= BLOCKFWD_MAX_KT && a.NT1 <= 16 && a.NT2 <= 16 && a.NT1 == 2 * a.KT1 &&
           (size_t)(a.KT0 + 2 * a.KT1) * 1024 <= 150 * 1024;
}
bool dec_bwd_rows_ok(const DecBwdRowsArgs& a) {
    return a.KT <= BLOCKFWD_MAX_KT && a.NT1 <= 16 && a.NT1 == 2 * a.KT && a.NT3 <= 16 && (size_t)(a.KTX + 2 * a.KT) * 1024 <= 150 * 1024;
}
void launch_dec_bwd_rows(const DecBwdRowsArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(a.KTX + 2 * a.KT) * 1024 + 64;      // (+ the 16 row weights of lse_on)
    if (a.M <= 512 && a.KTX > 8) LAUNCH_EV((dec_bwd_rows_kernel<13>), dim3((a.M + 15) / 16), dim3(1024), lds, st, a);
    else LAUNCH_EV((dec_bwd_rows_kernel<6>), dim3((a.M + 15) / 16), dim3(1024), lds, st, a);
}
bool block_bwd_ok(const BlockBwdArgs& a) {
    return a.R <= 4096 && a.KTH <= BLOCKFWD_MAX_KT && a.KT1 <= BLOCKFWD_MAX_KT && a.NT1 <= 16 && a.NT1 == 2 * a.KT1;
}
void launch_block_bwd(const BlockBwdArgs& a, hipStream_t st) {
    if (a.rows_per_wg == 4) hipLaunchKernelGGL(block_bwd_kernel<4>, dim3((a.R + 3) / 4), dim3(1024), (size_t)(a.KTH + a.KT1) * 1024 + 16384, st, a);
    else hipLaunchKernelGGL(block_bwd_kernel<16>, dim3((a.R + 15) / 16), dim3(1024), (size_t)(a.KTH + a.KT1) * 1024, st, a);
}
void launch_block_fwd(const BlockFwdArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(a.KT0 + 2 * a.KT1) * 1024 + (a.sample ? (size_t)a.KT0 * 768 : 0);
    // a handful of workgroups and a long first layer (the encoder on <= 512 images): 13 weight fragments in flight per wave instead of
    // 6 (the 25 k-steps of the 784-pixel layer in two round trips to L2 instead of five) -- at the price of the whole CU's registers
    if (a.oimg) { hipLaunchKernelGGL((block_fwd_kernel<6, true>), dim3((a.R + 15) / 16), dim3(1024), lds, st, a); return; }
    if (a.R <= 512 && a.KT0 > 8) hipLaunchKernelGGL((block_fwd_kernel<13>), dim3((a.R + 15) / 16), dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((block_fwd_kernel<6>), dim3((a.R + 15) / 16), dim3(1024), lds, st, a);
}
// the pipelined Bernoulli forward exists for the reference's hidden width (7 k-steps), one block owning all pixel groups,
// and k large enough that a block's 128 rows span <= BERN_XIMG_MAX images
bool bern_pipe_ok(const DenseArgs& a) {
    if (a.pre_img1 && (a.pre_KT1 < 1 || a.pre_KT1 > 4)) return false;
    return a.KT == 7 && a.M >= 8192 && a.mg_per_block >= a.MG && !a.logits_out && a.lpxz_stride == 0 &&
           (126 + a.k) / a.k + 1 <= BERN_XIMG_MAX && (a.Np32 >> 5) >= 2 && (a.Np32 >> 5) <= 2 * a.MG &&
           2 * (7 * 4096 + 1024) + (size_t)BERN_XIMG_MAX * a.ldXB * 4 + 128 <= 80 * 1024;
}
// the decoder kernel can finish its images' log-mean-exp itself: 16-wave / 200-row shape with the whole decoder inside, k a divisor of 200 with
// at most 16 images per workgroup (k = 20, 25, 40, 50, 100, 200), terms 1 and 2 the ones the kernel makes when it samples z itself
bool bern_lse_ok(const DenseArgs& a) {
    return bern_pipe_ok(a) && a.pre_img1 && a.pipe >= 2 && (198 + a.k) / a.k + 1 <= 8 && a.k >= 13 && 200 % a.k == 0 &&
           (!a.zhead || (a.lse.term[1] == a.zlp && a.lse.term[2] == a.zlq && (!a.lse.lq_dreg || a.lse.lq_dreg == a.zlq_dreg))) && a.lse.n_px_part <= 1;
}
void launch_dense(int epi, const DenseArgs& a, hipStream_t st) {
    dim3 grid((a.M + 127) / 128, (a.MG + a.mg_per_block - 1) / a.mg_per_block);
    const size_t lds = 2 * DENSE_UNIT;
    if (epi == EPI_BERN && a.pipe && bern_pipe_ok(a)) {
        const size_t ldsb = 2 * (7 * 4096 + 1024) + (size_t)BERN_XIMG_MAX * a.ldXB * 4 + 128;
        if (a.pre_img1 && a.pipe >= 2 && (198 + a.k) / a.k + 1 <= 8) {       // 16-wave / 200-row shape (see QW)
            const size_t ldsq = 2 * (7 * 4096 + 1024) + (size_t)8 * a.ldXB * 4 + 128 + 4096 + 2 * 7 * 1024 + 4096 + 1024      // (last 5 KiB: the rows' terms for the in-kernel lse_image, 208 floats each, and its row weights)
                                + (a.zhead ? (size_t)8 * 3 * a.zDp * 4 + 256 : 0);      // (the sampling prologue's heads of <= 8 images: mu | sigma | sigma/(sigma + 1e-6), + their log-sigma sums)
            if (a.YP) LAUNCH_EV((bern_pipe_kernel<7, true, true, true>), dim3((a.M + 199) / 200), dim3(1024), ldsq, st, a);
            else LAUNCH_EV((bern_pipe_kernel<7, false, true, true>), dim3((a.M + 199) / 200), dim3(1024), ldsq, st, a);
        } else if (a.pre_img1) {       // the whole decoder in one launch
            if (a.YP) LAUNCH_EV((bern_pipe_kernel<7, true, true>), dim3(grid.x), dim3(512), ldsb, st, a);
            else LAUNCH_EV((bern_pipe_kernel<7, false, true>), dim3(grid.x), dim3(512), ldsb, st, a);
        } else if (a.YP) LAUNCH_EV((bern_pipe_kernel<7, true, false>), dim3(grid.x), dim3(512), ldsb, st, a);
        else LAUNCH_EV((bern_pipe_kernel<7, false, false>), dim3(grid.x), dim3(512), ldsb, st, a);
        return;
    }
    if (launch_dense_g1(epi, a, grid, lds, st)) return;
    // compile-time k-step counts for the reference model's shapes (200->224, 100->128, 50->64, head 256)
    switch (a.KT) {
        case 2: launch_dense_k<2>(epi, a, grid, lds, st); break;
        case 4: launch_dense_k<4>(epi, a, grid, lds, st); break;
        case 7: launch_dense_k<7>(epi, a, grid, lds, st); break;
        case 8: launch_dense_k<8>(epi, a, grid, lds, st); break;
        default: launch_dense_k<0>(epi, a, grid, a.stage_all ? 4 * DENSE_UNIT : lds, st); break;
    }
}
bool chain2_fwd_ok(int KT0, int KTH, int KT1, int M) { return KT0 == 4 && KTH == 4 && KT1 == 2 && M >= 8192; }
void launch_chain2_fwd(const Chain2FwdArgs& a, hipStream_t st) {
    LAUNCH_EV((chain2_fwd_kernel<4, 4, 2>), dim3((a.M + 127) / 128), dim3(512), 2 * (4 * 4096 + 1024), st, a);
}
bool gblock_bwd_ok(int KT0, int KTH, int KT1, int M) { return KT0 == 4 && KTH == 4 && KT1 == 2 && M >= 8192; }
void launch_gblock_bwd(int mode, const GBlockBwdArgs& a, hipStream_t st) {
    dim3 grid((a.M + 127) / 128);
    if (mode == 0) LAUNCH_EV((gblock_bwd_kernel<0, 2, 4, 4>), grid, dim3(512), 2 * (8 * 4096 + 1024), st, a);      // decode_z2_to_z1: head over z1 (KTL = 4), input z2
    else LAUNCH_EV((gblock_bwd_kernel<1, 4, 4, 2>), grid, dim3(512), 2 * (4 * 4096 + 1024), st, a);                // encode_z1_to_z2: head over z2 (KTL = 2), input z1
}
bool out_bwd_has_s_mode(int KT) { return KT == 7 || KT == 4 || KT == 2; }
void launch_dec_bwd(const DecBwdArgs& d, hipStream_t st) {
    const size_t lds = 2 * ((size_t)d.o.KT * 4096 + 1024);
    dim3 grid((d.o.M + 127) / 128);
    if (d.nw == 8 && d.o.KT == 7) { LAUNCH_EV((dec_bwd_kernel<7, 8, 1>), grid, dim3(512), lds, st, d); return; }
    switch (d.o.KT) {
        case 7: LAUNCH_EV(dec_bwd_kernel<7>, grid, dim3(256), lds, st, d); break;
        case 4: LAUNCH_EV(dec_bwd_kernel<4>, grid, dim3(256), lds, st, d); break;
        case 2: LAUNCH_EV(dec_bwd_kernel<2>, grid, dim3(256), lds, st, d); break;
        default: break;      // the host only asks where out_bwd_has_s_mode(KT)
    }
}
void launch_out_bwd(const OutBwdArgs& a, hipStream_t st) {
    if (a.SP) {      // the forward pass kept s = x - sigmoid(l): one product, B operand from HBM
        const size_t lds = 2 * ((size_t)a.KT * 4096 + 1024);
        const int nparts = a.part ? (a.NG + a.gpb - 1) / a.gpb : 1;
        dim3 grid((a.M + 127) / 128, nparts);
        hipEvent_t stop = g_stop_event;      // belongs to the LAST kernel of this wrapper
        if (a.part) g_stop_event = nullptr;
        switch (a.KT) {
            case 7: LAUNCH_EV(out_bwd_s_kernel<7>, grid, dim3(256), lds, st, a); break;
            case 4: LAUNCH_EV(out_bwd_s_kernel<4>, grid, dim3(256), lds, st, a); break;
            case 2: LAUNCH_EV(out_bwd_s_kernel<2>, grid, dim3(256), lds, st, a); break;
            default: return;      // the host only asks for this mode when out_bwd_has_s_mode(KT)
        }
        if (a.part) {
            g_stop_event = stop;
            LAUNCH_EV(out_bwd_finish_kernel, grid1((size_t)a.M * (a.ldG / 8), 256), dim3(256), 0, st, a, nparts);
        }
        return;
    }
    // pair kernel: one W^T image per pixel group, double buffered, + 4 KiB of dl exchange per pair.  Two 4-wave
    // workgroups (64 rows each) per CU instead of one 8-wave workgroup: the two are not in lockstep, so one's
    // MFMA phase overlaps the other's sigmoid epilogue / DMA.
    constexpr int PAIRS = 2;
    const size_t ldsp = 2 * ((size_t)a.KT * 4096 + 1024) + (size_t)PAIRS * 4096;
    dim3 gridp((a.M + PAIRS * 32 - 1) / (PAIRS * 32));
#ifdef IWAE_DIAG
    if (a.stamps) {   // diagnostic build
        LAUNCH_EV((out_bwd_pair_kernel<7, PAIRS, true>), gridp, dim3(PAIRS * 128), 2 * ((size_t)7 * 4096 + 1024) + (size_t)PAIRS * 4096, st, a);
        return;
    }
#endif
    switch (a.KT) {
        case 7: LAUNCH_EV((out_bwd_pair_kernel<7, PAIRS, false>), gridp, dim3(PAIRS * 128), ldsp, st, a); break;
        case 4: LAUNCH_EV((out_bwd_pair_kernel<4, PAIRS, false>), gridp, dim3(PAIRS * 128), ldsp, st, a); break;
        case 2: LAUNCH_EV((out_bwd_pair_kernel<2, PAIRS, false>), gridp, dim3(PAIRS * 128), ldsp, st, a); break;
        default: {   // run-time hidden width: generic 4-wave kernel with both weight images
            const size_t lds = 2 * ((size_t)a.KT * 8192 + 1024);
            LAUNCH_EV((out_bwd_kernel<0, false>), dim3((a.M + 127) / 128), dim3(256), lds, st, a);
        }
    }
}
void launch_wgradws_group(const WgradPGroup& g, hipStream_t st) {
    int mx = 0;
    for (int l = 0; l < g.n; ++l) mx = std::max(mx, g.gx[l]);
    LAUNCH_EV(wgradws_group_kernel, dim3(mx, 1, g.zbeg[g.n]), dim3(768), WG_NST * (WG_SR * 512 + WG_SR * 512), st, g);
}
void launch_wgradp_group(const WgradPGroup& g, hipStream_t st) {
    int mx = 0, my = 0;
    for (int l = 0; l < g.n; ++l) { mx = std::max(mx, g.gx[l]); my = std::max(my, g.gy[l]); }
    if (g.a[0].rowscale) hipLaunchKernelGGL(wgradp_group_sc_kernel, dim3(mx, my, g.zbeg[g.n]), dim3(512), WG_NST * (WG_SR * 512 + WG_SR * 256 + 1024), st, g);
    else hipLaunchKernelGGL(wgradp_group_kernel, dim3(mx, my, g.zbeg[g.n]), dim3(512), WG_NST * (WG_SR * 512 + WG_SR * 256), st, g);
}
// shape: 8 = 8 waves / 8 j-tiles per block (small), 16 = 16 waves / 16 j-tiles; specialised waves (IT <= 14): 7 = 8 + 4 waves / 16 j-tiles,
// 9 = 8 + 8 waves / 8 j-tiles
int wgradp_strip(int shape) { return (shape == 8 || shape == 9) ? 8 : 16; }
void launch_wgradp(const WgradPArgs& a, int nsplit, int shape, hipStream_t st) {
    dim3 grid((a.JT + wgradp_strip(shape) - 1) / wgradp_strip(shape), (a.IT + 15) / 16, nsplit);
    const size_t sc = a.rowscale ? 1024 : 0;
    if (shape == 7) {        // specialised waves: 8 compute + 4 loader (needs IT <= 14, one i-block)
        const size_t lds = WG_NST * (WG_SR * 512 + WG_SR * 512);
        if (a.rowscale) LAUNCH_EV((wgradws_kernel<true, 4, 4>), grid, dim3(768), lds, st, a);
        else LAUNCH_EV((wgradws_kernel<false, 4, 4>), grid, dim3(768), lds, st, a);
    } else if (shape == 9) { // specialised waves: 8 compute + 8 loader, 128-feature column blocks
        const size_t lds = WG_NST * (WG_SR * 512 + WG_SR * 256);
        if (a.rowscale) LAUNCH_EV((wgradws_kernel<true, 2, 8>), grid, dim3(1024), lds, st, a);
        else LAUNCH_EV((wgradws_kernel<false, 2, 8>), grid, dim3(1024), lds, st, a);
    } else if (shape == 16) {
        const size_t lds = WG_NST * (WG_SR * 512 + WG_SR * 512 + sc);
        if (a.rowscale) LAUNCH_EV((wgradp_kernel<16, 2, 8, 2, true>), grid, dim3(1024), lds, st, a);
        else LAUNCH_EV((wgradp_kernel<16, 4, 4, 4, false>), grid, dim3(1024), lds, st, a);
    } else {
        const size_t lds = WG_NST * (WG_SR * 512 + WG_SR * 256 + sc);
        if (a.rowscale) LAUNCH_EV((wgradp_kernel<8, 4, 4, 4, true>), grid, dim3(512), lds, st, a);
        else LAUNCH_EV((wgradp_kernel<8, 4, 4, 4, false>), grid, dim3(512), lds, st, a);
    }
}
void launch_prep_rows(const float* x, const float* cond, int B, int X, int C, int Xp, int Bp, uint16_t* XP, hipStream_t st) {
    const int nchunk = Xp / 8;
    hipLaunchKernelGGL(prep_rows_kernel, dim3(Bp / 64, std::min(32, (nchunk + 3) / 4)), dim3(256), 0, st, x, cond, B, X, cond ? C : 0, Xp, Bp, XP);
}
void launch_gather_binarize(const uint8_t* data, const int32_t* order, int start, int N, int B, int X, int Xp, int Bp, uint64_t seed,
                            uint32_t epoch, uint16_t* XP, float* xf, hipStream_t st, const uint8_t* labels, int C, float* cond_out) {
    const int nchunk = Xp / 8;
    hipLaunchKernelGGL(gather_binarize_kernel, dim3(Bp / 64, std::min(32, (nchunk + 3) / 4)), dim3(256), 0, st, data, order, start, N, B, X, Xp,
                       Bp, seed, epoch, XP, xf, labels, labels ? C : 0, cond_out);
}
void launch_eps_gen_multi(const EpsSrc& e, int M, int D, int ld, float* out, int nsteps, size_t step_stride, hipStream_t st) {
    const int nd4 = (D + 3) / 4;
    hipLaunchKernelGGL(eps_gen_multi_kernel, grid1((size_t)nsteps * M * nd4, 256), dim3(256), 0, st, e, M, nd4, ld, out, nsteps, step_stride);
}
void launch_eps_gen(const EpsSrc& e, int M, int D, int ld, float* out, hipStream_t st, int max_blocks) {
    const int nd4 = (D + 3) / 4;
    dim3 grid = grid1((size_t)M * nd4, 256);
    if (max_blocks > 0 && (int)grid.x > max_blocks) grid.x = max_blocks;
    hipLaunchKernelGGL(eps_gen_kernel, grid, dim3(256), 0, st, e, M, nd4, ld, out);
}
void launch_sample(const SampleArgs& a, hipStream_t st) { hipLaunchKernelGGL(sample_kernel, dim3((a.M + 63) / 64), dim3(256), 0, st, a); }
void launch_gauss_lp(const GaussLpArgs& a, hipStream_t st) { hipLaunchKernelGGL(gauss_lp_kernel, dim3((a.M + 63) / 64), dim3(256), 0, st, a); }
void launch_lse(const LseArgs& a, hipStream_t st) {
    if (a.k >= 2048) LAUNCH_EV(lse_block_kernel<16>, dim3(a.B), dim3(1024), 0, st, a);
    else if (a.k > 256) LAUNCH_EV(lse_block_kernel<4>, dim3(a.B), dim3(256), 0, st, a);
    else LAUNCH_EV(lse_kernel, grid1((size_t)a.B * 64, 256), dim3(256), 0, st, a);
}
void launch_latent_bwd(const LatentBwdArgs& a, hipStream_t st) {
    if (a.dzh && !a.dz2 && a.eps.cache && !a.prior_head) hipLaunchKernelGGL(latent_bwd_kernel<1>, dim3(a.Bp), dim3(256), 0, st, a);
    else if (!a.dzh && a.dz && a.dz2 && a.dz3 && a.eps.cache && !a.prior_head) hipLaunchKernelGGL(latent_bwd_kernel<2>, dim3(a.Bp), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(latent_bwd_kernel<0>, dim3(a.Bp), dim3(256), 0, st, a);
}
void launch_gauss_bwd(const GaussBwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(gauss_bwd_kernel, grid1((size_t)a.Mp * (a.Dp / 4), 256), dim3(256), 0, st, a);
}
void launch_reduce_grads(const LayerDesc* layers, int nlayers, int first_block, int nblocks, float* grad, float* param, float* mom, float* vel,
                         float alpha, float beta1, float beta2, float eps, int fuse_adam, const float* per_b, int B, float beta, float* scalars, hipStream_t st,
                         int first_block2, int nblocks2) {
    const AdamCoef c = {alpha, 1.0f, beta1, beta2, eps, fuse_adam};
    const MeansArgs mn = {per_b, B, beta, scalars};
    LAUNCH_EV(reduce_grads_kernel, dim3(nblocks + nblocks2 + (per_b ? 1 : 0)), dim3(256), 0, st, layers, nlayers, grad, param, mom, vel, c, mn, first_block, nblocks,
              first_block2);
}
void launch_scalars(const float* per_b, int B, float beta, float* out, hipStream_t st) {
    hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(256), 0, st, per_b, B, beta, out);
}
void launch_adam(const LayerDesc* layers, int nlayers, int nblocks, float* param, const float* grad, float* mom, float* vel,
                 float alpha, float gscale, float beta1, float beta2, float eps, int do_update, hipStream_t st, int first_block) {
    const AdamCoef c = {alpha, gscale, beta1, beta2, eps, do_update};
    LAUNCH_EV(adam_kernel, dim3(nblocks), dim3(256), 0, st, layers, nlayers, param, grad, mom, vel, c, first_block);
}
void launch_wgrad_rows(const WgradRowsJob* jobs, int njobs, const LayerDesc* layers, float* grad, float* param, float* mom, float* vel, float alpha,
                       float beta1, float beta2, float eps, int fuse_adam, const float* per_b, int B, float beta, float* scalars, const char* zero, hipStream_t st) {
    WgradRowsArgs a;
    memset(&a, 0, sizeof(a));
    int wgs = 0;
    for (int i = 0; i < njobs && i < WGR_MAX_JOBS; ++i) {
        a.job[i] = jobs[i];
        a.job[i].ib = (jobs[i].ldX + 63) / 64;
        a.job[i].jb = (jobs[i].ldG + 63) / 64;
        a.job[i].wg_begin = wgs;
        wgs += a.job[i].ib * a.job[i].jb;
    }
    a.njobs = njobs;
    a.layers = layers; a.grad = grad; a.param = param; a.mom = mom; a.vel = vel;
    a.c = AdamCoef{alpha, 1.0f, beta1, beta2, eps, fuse_adam};
    a.mn = MeansArgs{per_b, B, beta, scalars};
    a.zero = zero;
    LAUNCH_EV(wgrad_rows_kernel, dim3(wgs + (per_b ? 1 : 0)), dim3(256), (size_t)4 * WGR_NST * WGR_STAGE, st, a);
}
void launch_export_rows(const float* in, int B, int k, float* out, hipStream_t st) {
    hipLaunchKernelGGL(export_rows_kernel, grid1((size_t)B * k, 256), dim3(256), 0, st, in, B, k, out);
}
void launch_export_z(const SampleArgs& a, float* zout, hipStream_t st) {
    hipLaunchKernelGGL(export_z_kernel, grid1((size_t)a.M * ((a.D + 3) / 4), 256), dim3(256), 0, st, a, zout, (const float*)nullptr, (float*)nullptr);
}
void launch_snis(const float* z, const float* wn, int B, int k, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(snis_kernel, grid1((size_t)B * D, 256), dim3(256), 0, st, z, wn, B, k, D, out);
}
void launch_unpack_p(const uint16_t* P, int rows, int F, int Fp, float* out, hipStream_t st) {
    hipLaunchKernelGGL(unpack_p_kernel, grid1((size_t)rows * F, 256), dim3(256), 0, st, P, rows, F, Fp, out);
}
void launch_eps_dump(const EpsSrc& e, int B, int k, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(eps_dump_kernel, grid1((size_t)B * k * ((D + 3) / 4), 256), dim3(256), 0, st, e, B, k, D, out);
}

}  // namespace iwae
