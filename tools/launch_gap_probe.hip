// GPU-box probe: what does a kernel BOUNDARY cost inside one stream?  A chain of N dependent launches of a kernel that keeps every CU busy
// for ~T us, timed with events, against N x the kernel alone -- as plain launches, as a captured hipGraph, and (floor) as launches without
// the barrier bit (hipExtAnyOrderLaunch: valid only for independent kernels).  The frame of td2-psp34 @720x960 fp16 is 58 launches in
// 0.92 ms, the fp32 headline frame 90 launches (+ events between three streams) in 3.6 ms.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_gap_probe.hip -o tools/_build/launch_gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_busy(float* p, unsigned long long ticks, int n) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    float v = p[(blockIdx.x * blockDim.x + threadIdx.x) % n];
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) v = v * 1.0001f + 0.5f;
    p[(blockIdx.x * blockDim.x + threadIdx.x) % n] = v;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    float* d; const int n = 1 << 20;
    CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int us : {0, 5, 15, 50}) {
        for (int grid : {64, 256, 1024}) {
            const unsigned long long ticks = (unsigned long long)us * 100ull;
            auto plain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_busy, dim3(grid), dim3(256), 0, s, d, ticks, n); };
            auto anyorder = [&]() { for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(k_busy, dim3(grid), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, ticks, n); };
            auto timeit = [&](auto&& f) { f(); CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s)); f(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / N; };
            const float t_plain = timeit(plain);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            plain();
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            auto graph = [&]() { CK(hipGraphLaunch(ge, s)); };
            const float t_graph = timeit(graph);
            const float t_any = timeit(anyorder);
            printf("kernel ~%2d us x %4d workgroups: per launch %6.2f us plain, %6.2f us in a hipGraph, %6.2f us without the barrier bit (independent)\n", us, grid, t_plain, t_graph, t_any);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
