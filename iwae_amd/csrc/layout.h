// Data layouts shared by the HIP kernels and the host side (gfx950, wave64).
//
// Everything is organised around ONE ownership rule that follows from the
// v_mfma_f32_16x16x32_bf16 register maps (cdna_hip_programming.md section 3):
//
//   a wave owns a block of data rows; lane (rho = lane & 15, q = lane >> 4) owns data
//   row rho of a 16-row column group, and of every 16 consecutive features the four
//   features 4q .. 4q+3.
//
// * accumulator tile (16 out-features x 16 rows): lane(rho,q) reg i = feature 16t'+4q+i
// * B operand of k-step s (32 in-features x 16 rows): lane(rho,q) element j =
//   feature 32s + 16(j>>2) + 4q + (j&3)  -- the SAME features the lane owns in the
//   accumulator tiles 2s (j<4) and 2s+1 (j>=4).  So a converted accumulator is the
//   next layer's B operand with no lane movement, and elementwise ops between a
//   stored activation and a fresh accumulator are lane-local.
//
// P-layout  (bf16 [rows][Fp], Fp = round_up(F,32)): features stored in ownership
//           order so a lane's 8 values of one k-step are 16 contiguous bytes:
//           position(f) = 32*(f/32) + 8*((f%16)/4) + 4*((f%32)/16) + f%4.
//           Every activation and activation-gradient lives in HBM in this one layout; the
//           weight-gradient GEMM (reduction over rows) transposes 4x4 blocks on the way out of
//           LDS with ds_read_b64_tr_b16 instead of keeping feature-major copies.
// A-image   (bf16 weights as the MFMA A operand, already in LDS order): 1 KiB blocks
//           of 16 out-features x 32 in-features; lane(rr,q) reads 16 B at
//           rr*64 + ((q ^ hp(rr>>2))*16), which is bank-conflict-free for
//           ds_read_b128's four 16-lane groups (MI355X_MICROARCH.md, LDS table).
//           Blocks are ordered [mgroup(64 out)][kstep][tile(4)] + one 1 KiB block per mgroup
//           holding the fp32 bias of its 64 out-features ("MG-major", streamed one
//           out-feature group at a time) or [kgroup(64 in)][kk(2)][mtile]
//           ("K-major", streamed one in-feature group at a time).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define IWAE_HD __host__ __device__ __forceinline__
#else
#define IWAE_HD inline
#endif

namespace iwae {

IWAE_HD int round_up(int x, int m) { return (x + m - 1) / m * m; }
IWAE_HD int hperm(int x) { return (0x78 >> (2 * x)) & 3; }   // {0,2,3,1}

// position of feature f inside a P-layout row
IWAE_HD int p_pos(int f) { return (f & ~31) + 8 * ((f & 15) >> 2) + 4 * ((f & 31) >> 4) + (f & 3); }

// byte offset of element (out-feature m, in-feature kf) inside an MG-major image
// whose k extent is KT k-steps (KT = round_up(K,32)/32)
IWAE_HD size_t img_mg_group_bytes(int KT) { return (size_t)(4 * KT + 1) * 1024; }
IWAE_HD size_t img_mg_byte(int m, int kf, int KT) {
    const int mg = m >> 6, tile = (m >> 4) & 3, rr = m & 15;
    const int ks = kf >> 5, h = (kf >> 4) & 1, q = (kf >> 2) & 3, i = kf & 3;
    return (size_t)mg * img_mg_group_bytes(KT) + (size_t)(ks * 4 + tile) * 1024 + rr * 64 + ((q ^ hperm(rr >> 2)) * 16) + (4 * h + i) * 2;
}
// byte offset of the fp32 bias of out-feature m (last block of its group)
IWAE_HD size_t img_mg_bias_byte(int m, int KT) {
    return (size_t)(m >> 6) * img_mg_group_bytes(KT) + (size_t)KT * 4096 + (m & 63) * 4;
}

// byte offset inside a K-major image with MT out-feature tiles (MT = round_up(Mout,16)/16)
IWAE_HD size_t img_k_byte(int m, int kf, int MT) {
    const int kg = kf >> 6, kk = (kf >> 5) & 1, h = (kf >> 4) & 1, q = (kf >> 2) & 3, i = kf & 3;
    const int mt = m >> 4, rr = m & 15;
    return ((size_t)(kg * 2 + kk) * MT + mt) * 1024 + rr * 64 + ((q ^ hperm(rr >> 2)) * 16) + (4 * h + i) * 2;
}

}  // namespace iwae
