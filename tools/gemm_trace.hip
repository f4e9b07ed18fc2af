// tools/gemm_trace.hip -- GPU-box probe (not part of the product library): k_gemm_persistent compiled with TD_GEMM_TRACE, one launch,
// dumps per-workgroup timestamps (start, end of each tile's K loop, end of each tile's epilogue) + the CU/SIMD slot it ran on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTD_GEMM_TRACE -Iinclude -Itdnet_amd/csrc tools/gemm_trace.hip -o tools/_build/gemm_trace
//   tools/_build/gemm_trace M N K [role] > trace.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "td_device.h"
#include "td_gemm.h"

__global__ void k_fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 131072, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 128;
    const int nbatch = argc > 4 ? atoi(argv[4]) : 1;
    const int tile = argc > 5 ? atoi(argv[5]) : 0, dynamic = argc > 7 ? atoi(argv[7]) : 0;   // tile 0: 128x128, 1: 64x128 (argv[6] was `stagger`, removed with the field in round 5; the position is kept)
    float *a, *w, *bias, *out; unsigned long long* tr;
    const size_t na = (size_t)nbatch * M * K, nw = (size_t)nbatch * K * N, no = (size_t)nbatch * M * N;
    hipMalloc(&a, na * 4); hipMalloc(&w, nw * 4); hipMalloc(&bias, N * 4); hipMalloc(&out, no * 4);
    const int grid = tile ? 768 : 512, BM = tile ? 64 : 128;
    hipMalloc(&tr, grid * 64 * 8); hipMemset(tr, 0, grid * 64 * 8);
    k_fill<<<1024, 256>>>(a, na, 1); k_fill<<<1024, 256>>>(w, nw, 2); k_fill<<<1, 256>>>(bias, N, 3);
    GemmArgs g{};
    g.a = a; g.wp = w; g.bias = bias; g.resid = nullptr; g.out = out; g.M = M; g.N = N; g.NPad = N; g.K = K; g.nbatch = nbatch; g.act = 1;
    g.MP = M; g.trace = tr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {                                  // the last launch is the one dumped
        hipEventRecord(e0, 0);
        hipMemsetAsync(tr, 0, grid * 64 * 8, 0);
        gemm_launch(g, tile ? CT_64x128 : CT_128x128, 0, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 64);
    hipMemcpy(h.data(), tr, grid * 64 * 8, hipMemcpyDeviceToHost);
    const int tiles = (M / BM) * (N / 128) * nbatch, per_wg = (tiles + grid - 1) / grid;
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < grid; ++b) t0 = std::min(t0, h[b * 64 + 2]);
    printf("# dynamic %d tile %d M %d N %d K %d nbatch %d: %.4f ms, %d tiles, %d per workgroup; times in us from the first workgroup's start\n", dynamic, tile, M, N, K, nbatch, ms, tiles, per_wg);
    printf("# wg xcc se sh cu simd wave | start | (kloop_end epi_end) per tile\n");
    for (int b = 0; b < grid; ++b) {
        const unsigned long long* r = &h[b * 64];
        const unsigned hw = (unsigned)r[0];
        printf("%3d %u %u %u %2u %u %u | %7.2f |", b, (unsigned)r[1] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15,
               (double)(r[2] - t0) / 100.0);
        for (int t = 3; t < 64 && r[t] != 0; ++t) printf(" %7.2f", (double)(r[t] - t0) / 100.0);
        printf("\n");
    }
    return 0;
}
