// Developer probe (GPU box): what v_permlane16_swap_b32 does to two registers (hipcc --offload-arch=gfx950 -O2 permlane16_swap.hip -o /tmp/pls && /tmp/pls)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
    unsigned a = p[threadIdx.x], b = p[threadIdx.x + 64];
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x + 128] = r[0];
    p[threadIdx.x + 192] = r[1];
}
int main() {
    unsigned h[256], *d;
    for (int i = 0; i < 64; ++i) { h[i] = 100 + i; h[64 + i] = 200 + i; }
    hipMalloc(&d, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("first result (old a = 100 + lane, old b = 200 + lane), one value per 16-lane row start and row end:\n");
    for (int r = 0; r < 4; ++r) printf("  a' row %d: %u .. %u     b' row %d: %u .. %u\n", r, h[128 + 16 * r], h[128 + 16 * r + 15], r, h[192 + 16 * r], h[192 + 16 * r + 15]);
    return 0;
}
