// Developer probe (GPU box): what a cross-stream dependency costs on each side.
//   s1: A -> [record e] -> B          s2: [wait e] -> C
// Every kernel stamps s_memrealtime (100 MHz) at start and end.  Reported: B.start - A.end with and without the record
// (the bubble an event record puts into its own stream), C.start - A.end (how long the waiting stream takes to go).
// hipcc --offload-arch=gfx950 -O2 -o event_latency event_latency.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long* out, int slot, int ticks) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[2 * slot] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
}
int main() {
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e, e2;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    unsigned long long h[64];
    const int T = 1000;     // 10 us
    unsigned int* flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
    unsigned int fval = 0;
    for (int mode = 0; mode < 7; ++mode) {
        std::vector<double> ab, ac, ad;
        for (int it = 0; it < 60; ++it) {
            hipMemsetAsync(d, 0, 64 * 8, s1); hipStreamSynchronize(s1);
            // mode 0: plain A,B on s1.  1: record between.  2: record + s2 waits and runs C.  3: as 2, and s1 then WAITS for an event of s2 before D
            if (mode <= 3) {
            spin<<<256, 64, 0, s1>>>(d, 0, T);
            if (mode >= 1) hipEventRecord(e, s1);
            spin<<<256, 64, 0, s1>>>(d, 1, T);
            if (mode >= 2) { hipStreamWaitEvent(s2, e, 0); spin<<<256, 64, 0, s2>>>(d, 2, T / 2); }
            if (mode >= 3) { hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); spin<<<256, 64, 0, s1>>>(d, 3, T); }
            } else if (mode == 4 || mode == 5) {
                // 4: the event rides on A's own dispatch packet (hipExtLaunchKernelGGL stop event), s2 waits for it
                // 5: same + join back through an event that rides on C
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, nullptr, e, 0, d, 0, T);
                spin<<<256, 64, 0, s1>>>(d, 1, T);
                hipStreamWaitEvent(s2, e, 0);
                if (mode == 4) spin<<<256, 64, 0, s2>>>(d, 2, T / 2);
                else {
                    hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s2, nullptr, e2, 0, d, 2, T / 2);
                    hipStreamWaitEvent(s1, e2, 0); spin<<<256, 64, 0, s1>>>(d, 3, T);
                }
            } else {
                // 6: stream memory operations: s1 writes a value behind A, s2 waits for it
                ++fval;
                spin<<<256, 64, 0, s1>>>(d, 0, T);
                hipStreamWriteValue32(s1, flag, fval, 0);
                spin<<<256, 64, 0, s1>>>(d, 1, T);
                hipStreamWaitValue32(s2, flag, fval, hipStreamWaitValueGte, 0xffffffffu);
                spin<<<256, 64, 0, s2>>>(d, 2, T / 2);
            }
            { hipError_t er = hipDeviceSynchronize(); if (er != hipSuccess) { printf("mode %d: %s\n", mode, hipGetErrorString(er)); return 1; } }
            { hipError_t er = hipGetLastError(); if (er != hipSuccess) { printf("mode %d launch: %s\n", mode, hipGetErrorString(er)); return 1; } }
            hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
            ab.push_back((double)(h[2] - h[1]) / 100.0);
            if (mode >= 2 || mode == 6) ac.push_back((double)(h[4] - h[1]) / 100.0);
            if (mode == 3 || mode == 5) ad.push_back((double)(h[6] - std::max(h[3], h[5])) / 100.0);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        fflush(stdout); printf("mode %d: B.start - A.end %.2f us   C.start - A.end %.2f us   D.start - max(B,C).end %.2f us\n", mode, med(ab), med(ac), med(ad));
    }
    return 0;
}
