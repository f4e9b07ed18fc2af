// GPU-box probe: what does ONE CU get out of the global -> LDS DMA path, against plain vector loads?  One workgroup per CU (256 of
// them), NW waves, every wave streams 1 KB pieces of an L2-resident 64 KB region of its workgroup for N rounds with a bounded number
// of pieces in flight.  Modes: 0 = buffer_load_dwordx4 ... lds (16 B per lane), 1 = buffer_load_dword ... lds (4 B per lane),
// 2 = buffer_load_dwordx4 into VGPRs, 3 = the same + ds_write_b128, 4 = half the pieces by DMA and half through VGPRs + ds_write.
// Prints bytes per shader clock and CU (s_memtime of workgroup 0) and the aggregate rate.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/lds_dma_bw.hip -o tools/_build/lds_dma_bw
#include <hip/hip_runtime.h>
#include "../tdnet_amd/csrc/td_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long g_cycles[2];

template <int MODE, int NW>
__global__ void __launch_bounds__(64 * NW, 1) k_stream(const float* src, int rounds, float* sink) {
    TD_DYN_LDS(smem);
    const int lane = threadIdx.x & 63, wave = td_wave();
    const TdBuf buf = td_make_buf(src + (size_t)blockIdx.x * 16384, 65536u);   // 64 KB per workgroup
    char* my = smem + wave * 8 * 1024;                                          // 8 KB of LDS per wave: a ring of eight pieces
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = (unsigned)(((r * 8 + j) * NW + wave) & 63) * 1024u + (unsigned)lane * 16u;
            if (MODE == 0 || (MODE == 4 && (j & 1) == 0)) td_buf_ld16_lds(buf, my + j * 1024, off, 0u);
            else if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(buf.r, (__attribute__((address_space(3))) void*)(my + j * 1024 + q * 256), 4,
                                                             (unsigned)(((r * 8 + j) * NW + wave) & 63) * 1024u + q * 256u + lane * 4u, 0u, 0, 0);
            } else {
                const f32x4 v = td_buf_ld4(buf, off, 0u);
                if (MODE == 2) keep = keep + v;
                else *reinterpret_cast<f32x4*>(my + j * 1024 + lane * 16) = v;
            }
        }
        if (MODE == 0 || MODE == 1) TD_WAIT_VM_PIECES(8);                       // about one round in flight
    }
    TD_WAIT_VM_PIECES(0);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_cycles[0] = __builtin_amdgcn_s_memtime() - t0; }
    if (MODE != 2) keep = *reinterpret_cast<f32x4*>(my + lane * 16);
    if (keep[0] == 12345.678f) sink[threadIdx.x] = keep[1];
}

template <int MODE, int NW>
static void run(const char* what, const float* src, float* sink, int rounds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_stream<MODE, NW>), dim3(256), dim3(64 * NW), NW * 8192, 0, src, rounds, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_stream<MODE, NW>), dim3(256), dim3(64 * NW), NW * 8192, 0, src, rounds, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cycles), sizeof(c));
    const double bytes_cu = (double)rounds * 8 * 1024 * NW;
    printf("%-44s %d waves: %6.1f us  %5.1f B/clk/CU (s_memtime)  %6.2f TB/s aggregate, %s\n", what, NW, ms * 1e3, bytes_cu / (double)c[0],
           bytes_cu * 256 / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    float *src, *sink;
    hipMalloc(&src, 256 * 65536); hipMalloc(&sink, 4096);
    hipMemset(src, 0, 256 * 65536);
    run<2, 4>("buffer_load_dwordx4 -> VGPR", src, sink, rounds);
    run<2, 8>("buffer_load_dwordx4 -> VGPR", src, sink, rounds);
    run<3, 4>("buffer_load_dwordx4 -> VGPR -> ds_write_b128", src, sink, rounds);
    run<3, 8>("buffer_load_dwordx4 -> VGPR -> ds_write_b128", src, sink, rounds);
    run<0, 4>("buffer_load_dwordx4 ... lds (DMA, 16 B/lane)", src, sink, rounds);
    run<0, 8>("buffer_load_dwordx4 ... lds (DMA, 16 B/lane)", src, sink, rounds);
    run<0, 16>("buffer_load_dwordx4 ... lds (DMA, 16 B/lane)", src, sink, rounds);
    run<1, 4>("buffer_load_dword ... lds (DMA, 4 B/lane)", src, sink, rounds);
    run<1, 8>("buffer_load_dword ... lds (DMA, 4 B/lane)", src, sink, rounds);
    run<4, 4>("half DMA, half VGPR + ds_write", src, sink, rounds);
    run<4, 8>("half DMA, half VGPR + ds_write", src, sink, rounds);
    return 0;
}
