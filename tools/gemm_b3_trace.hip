// tools/gemm_b3_trace.hip -- GPU-box probe (not part of the product library): k_gemm_b3<1> compiled with TD_B3_TRACE, one launch of the 36 GEMMs of a
// Winograd conv, dumps per-wave s_memtime stamps of the first 48 K steps of workgroups 0..7 (step start, MFMAs issued, after the vmcnt wait, after the barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Itdnet_amd/csrc tools/gemm_b3_trace.hip -o tools/_build/gemm_b3_trace
//   tools/_build/gemm_b3_trace [M rows per plane] [N] [K] [grid] [dead-traffic flags]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_b3_trace;
#define TD_B3_TRACE g_b3_trace
#include "td_device.h"
#include "td_gemm_b3.h"

__global__ void k_fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
__global__ void k_fill16(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (unsigned short)(0x3c00u + (x & 0x3ff));               // finite bf16 bit patterns of mixed magnitude
    }
}
__global__ void k_setp(unsigned long long* p) { g_b3_trace = p; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 2048, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512, grid = argc > 4 ? atoi(argv[4]) : 0;
    const int dead = argc > 5 ? atoi(argv[5]) : 0;                    // 1: A pieces zero-filled (no memory traffic), 2: B pieces, 3: both
    const int nb = 36;
    float *a, *out; unsigned short* w; unsigned long long* tr;
    const size_t na = (size_t)nb * M * K, nw = (size_t)nb * gemm_b3_packed_bytes(K, N) / 2, no = (size_t)nb * M * N;
    hipMalloc(&a, na * 4); hipMalloc(&w, nw * 2); hipMalloc(&out, no * 4);
    const size_t ntr = (size_t)8 * 4 * 48 * 4;
    hipMalloc(&tr, ntr * 8); hipMemset(tr, 0, ntr * 8);
    k_setp<<<1, 1>>>(tr);
    k_fill<<<1024, 256>>>(a, na, 1); k_fill16<<<1024, 256>>>(w, nw, 2);
    GemmArgs g{};
    g.a = a; g.wp = (const float*)w; g.bias = nullptr; g.resid = nullptr; g.out = out; g.M = M; g.N = N; g.K = K; g.nbatch = nb; g.act = dead; g.MP = M;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {                                  // the last launch is the one dumped
        hipMemsetAsync(tr, 0, ntr * 8, 0);
        hipEventRecord(e0, 0);
        gemm_b3_launch(g, grid, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ntr);
    hipMemcpy(h.data(), tr, ntr * 8, hipMemcpyDeviceToHost);
    printf("# dead-traffic flags %d | k_gemm_b3<1> M %d N %d K %d x 36: %.4f ms = %.1f TFLOP/s; cycles (s_memtime) per K step of 48 bf16 MFMAs per wave (1536 pipe cycles): issue = start -> MFMAs issued, dma = vmcnt wait, bar = barrier wait\n",
           dead, M, N, K, ms, 2.0 * nb * M * (double)N * K / ms / 1e9);
    for (int b = 0; b < 8; ++b)
        for (int wv = 0; wv < 4; ++wv) {
            double si = 0, sd = 0, sb = 0, st = 0; int n = 0;
            const unsigned long long* r = &h[((size_t)b * 4 + wv) * 48 * 4];
            for (int s = 2; s + 1 < 48; ++s) {
                if (!r[s * 4] || !r[(s + 1) * 4]) break;
                si += (double)(r[s * 4 + 1] - r[s * 4]); sd += (double)(r[s * 4 + 2] - r[s * 4 + 1]); sb += (double)(r[s * 4 + 3] - r[s * 4 + 2]);
                st += (double)(r[(s + 1) * 4] - r[s * 4]); ++n;
            }
            if (n) printf("wg %d wave %d: %2d steps, per step: total %7.0f | issue %7.0f | dma %6.0f | bar %6.0f\n", b, wv, n, st / n, si / n, sd / n, sb / n);
        }
    // one wave in detail: the first 40 steps of workgroup 0, wave 0
    printf("# wg 0 wave 0, per step: issue dma bar (cycles)\n");
    const unsigned long long* r = &h[0];
    for (int s = 0; s < 40 && r[s * 4]; ++s) printf("%2d: %6llu %6llu %6llu\n", s, r[s * 4 + 1] - r[s * 4], r[s * 4 + 2] - r[s * 4 + 1], r[s * 4 + 3] - r[s * 4 + 2]);
    return 0;
}
