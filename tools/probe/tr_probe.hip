// Probe: semantics of ds_read_b64_tr_b16 (the hardware-transposing LDS read) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short v4s;
__global__ void k(const short* in, v4s* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, l16 = lane & 15;
  const short* p = lds + grp * 64 + (l16 >> 2) * 16 + (l16 & 3) * 4;     // row q' = l16>>2 (16 shorts per row), columns 4p..4p+3
  out[lane] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
}
int main() {
  short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  short* d; v4s* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * sizeof(v4s));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  v4s r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int g = l >> 4, i = l & 15;
    for (int e = 0; e < 4; ++e) if (r[l][e] != 64 * g + 16 * e + i) ++bad;
  }
  for (int l : {0, 1, 5, 17, 63}) printf("lane %2d: %d %d %d %d\n", l, r[l][0], r[l][1], r[l][2], r[l][3]);
  printf("expected lane i of group g = {64g+i, 64g+16+i, 64g+32+i, 64g+48+i}: mismatches %d\n", bad);
  return 0;
}
