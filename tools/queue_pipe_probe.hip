// GPU-box probe: do two HIP streams that own DIFFERENT hardware queues still serialise when one of them is dispatching a grid larger than
// the chip holds?  (Hypothesis: the queues of a process are dealt onto the compute micro-engine's few pipes; a pipe processes one dispatch at
// a time, so a kernel arriving on a queue of the SAME pipe as a persistent / oversubscribed grid waits for that grid's last workgroup to be
// dispatched.)  Base = the null stream (or the first created stream with an argument); X_i = highest-priority streams created one after
// the other and kept alive.  Per X_i: a BIG kernel (32768 workgroups of 64 threads x 4 us) on base and a 1-workgroup MARKER on X_i,
// started together; reports when the marker finished relative to BIG.
// Build: hipcc --offload-arch=gfx950 -O3 tools/queue_pipe_probe.hip -o tools/_build/queue_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
int main(int argc, char** argv) {
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t base = nullptr;
    if (argc > 1) CK(hipStreamCreateWithFlags(&base, hipStreamNonBlocking));
    const int prio = argc > 2 ? least : greatest;
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    printf("base = %s; X_i at priority %d\n", argc > 1 ? "created non-blocking stream" : "null stream", prio);
    std::vector<hipStream_t> xs;
    for (int i = 0; i < 12; ++i) {
        hipStream_t x;
        CK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, prio));
        xs.push_back(x);
        float best_marker = 1e9f, big = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, base));
            CK(hipStreamWaitEvent(x, e0, 0));
            hipLaunchKernelGGL(k_spin, dim3(32768), dim3(64), 0, base, 400ull);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, x, 100ull);
            CK(hipEventRecord(e1, x));
            CK(hipEventRecord(e2, base));
            CK(hipDeviceSynchronize());
            float a = 0, b = 0;
            CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2));
            if (a < best_marker) { best_marker = a; big = b; }
        }
        printf("X_%-2d (stream #%d of its class): marker done after %6.1f us, BIG after %6.1f us  -> %s\n", i, i, best_marker * 1e3f, big * 1e3f,
               best_marker > 0.6f * big ? "WAITED for BIG's dispatch" : "ran beside BIG");
    }
    return 0;
}
