// GPU-box probe: which HIP streams of a process really run kernels CONCURRENTLY with the null stream?  (A handle runs its two row-parity
// chains on the caller's stream and an internal one; with idle handles alive the frame sometimes ran at 0.6x -- tools/idle_handle_probe.py.)
// For K = 0 .. 11: K filler streams are created (and used once), then a stream X; a 200-us spin kernel (one workgroup) goes to the null
// stream and one to X at the same moment (X waits for an event recorded on the null stream just before); wall time of the pair
// ~200 us = concurrent, ~400 us = serialised.  Also for X created with the highest / lowest priority.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_overlap_probe.hip -o tools/_build/stream_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && threadIdx.x == 1024) *sink = 1;
}
static int pair_us(hipStream_t a, hipStream_t x, float* us) {
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, a));
        CK(hipStreamWaitEvent(x, e0, 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 20000ull, (int*)nullptr);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, x, 20000ull, (int*)nullptr);
        CK(hipEventRecord(e1, a));
        CK(hipEventRecord(e2, x));
        CK(hipStreamWaitEvent(a, e2, 0));
        CK(hipEventRecord(e1, a));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    *us = best * 1e3f;
    return 0;
}
int main(int argc, char** argv) {
    const bool legacy = argc > 1;                                           // any argument: pair with a created stream instead of the null stream
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t base = nullptr;
    if (legacy) CK(hipStreamCreateWithFlags(&base, hipStreamNonBlocking));
    printf("pair = (%s, X); priorities: least %d greatest %d\n", legacy ? "a non-blocking stream created first" : "null stream", least, greatest);
    std::vector<hipStream_t> fill;
    for (int K = 0; K <= 11; ++K) {
        hipStream_t xs[3];
        CK(hipStreamCreateWithFlags(&xs[0], hipStreamNonBlocking));
        CK(hipStreamCreateWithPriority(&xs[1], hipStreamNonBlocking, greatest));
        CK(hipStreamCreateWithPriority(&xs[2], hipStreamNonBlocking, least));
        float us[3];
        for (int i = 0; i < 3; ++i) if (pair_us(base, xs[i], &us[i])) return 1;
        printf("K = %2d filler streams alive: X normal %5.0f us %s | X highest %5.0f us %s | X lowest %5.0f us %s\n", K, us[0], us[0] > 300 ? "SERIAL" : "concurrent",
               us[1], us[1] > 300 ? "SERIAL" : "concurrent", us[2], us[2] > 300 ? "SERIAL" : "concurrent");
        for (int i = 0; i < 3; ++i) CK(hipStreamDestroy(xs[i]));
        hipStream_t f;
        CK(hipStreamCreateWithFlags(&f, hipStreamNonBlocking));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, f, 10ull, (int*)nullptr);
        CK(hipDeviceSynchronize());
        fill.push_back(f);
    }
    return 0;
}
