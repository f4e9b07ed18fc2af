// GPU-box probe: how many workgroups of 320 threads (5 waves) with ~128 VGPRs and 49.4 KB of LDS does a CU hold at once?  (The
// "rider wave" design -- a fifth, transform wave in every persistent-GEMM workgroup -- needs three: 15 waves per CU, (4,4,4,3) per SIMD
// at <= 128 registers.)  768 workgroups each stamp start / end (s_memrealtime) and HW_ID; the host counts how many started within the
// first 10 % of the kernel: 768 = three per CU resident together, 512 = two.
// Build: hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/_build/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NV>
__global__ void __launch_bounds__(320) k(unsigned long long* out, int iters) {
    extern __shared__ char smem[];
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % NV];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    smem[threadIdx.x] = (char)s;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = t1;
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        out[blockIdx.x * 4 + 3] = (unsigned long long)smem[5] + __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
}
template <int NV>
void run(const char* name, int threads) {
    const int G = 768;
    unsigned long long* d;
    hipMalloc(&d, G * 4 * 8);
    hipLaunchKernelGGL(k<NV>, dim3(G), dim3(threads), 49408, 0, d, 20000);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<NV>, dim3(G), dim3(threads), 49408, 0, d, 20000);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(G * 4);
    hipMemcpy(h.data(), d, G * 4 * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < G; ++i) { tmin = std::min(tmin, h[i * 4]); tmax = std::max(tmax, h[i * 4 + 1]); }
    int early = 0;
    for (int i = 0; i < G; ++i) early += (h[i * 4] - tmin) * 10 < (tmax - tmin);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k<NV>);
    printf("%s: %d threads, %d VGPRs: %s; kernel %.1f us; workgroups started in the first 10%%: %d of %d\n", name, threads, fa.numRegs,
           hipGetErrorString(e), (tmax - tmin) / 100.0, early, G);
    hipFree(d);
}
int main() {
    run<100>("~128 VGPR, 5 waves", 320);
    run<100>("~128 VGPR, 4 waves", 256);
    run<88>("~112 VGPR, 5 waves", 320);
    run<72>("~96 VGPR, 5 waves", 320);
    return 0;
}
