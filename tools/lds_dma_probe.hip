// GPU-box probe: what `buffer_load_dwordx4 ... lds` (td_device.h td_buf_ld16_lds) does on gfx950 --
//   (1) the destination is lds_base + 16 * lane (lane-linear) whatever the per-lane source offsets are;
//   (2) a lane whose source offset is out of the buffer's range WRITES ZEROS (it does not skip its LDS slot): the padding taps of
//       k_conv_dma_h (td_conv_hd.h) rely on it;
//   (3) a kernel may ask for 144 KB of dynamic LDS.
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o tools/_build/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* dst, int n, int big) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 4, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    char* base = smem + (big ? 140 * 1024 : 0);
    float4* mine = reinterpret_cast<float4*>(base + threadIdx.x * 16);
    *mine = make_float4(-7.f, -7.f, -7.f, -7.f);                       // sentinel
    __syncthreads();
    // source: lanes permuted (lane ^ 5), odd lanes out of range
    const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)((wave * 64 + (lane ^ 5)) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(base + wave * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    reinterpret_cast<float4*>(dst)[threadIdx.x] = *mine;
}
int main() {
    const int n = 256 * 4;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int big = 0; big < 2; ++big) {
        hipMemset(o, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), big ? 144 * 1024 : 4096, 0, d, o, n, big);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> g(n);
        hipMemcpy(g.data(), o, n * 4, hipMemcpyDeviceToHost);
        int lin = 0, zeros = 0, kept = 0, other = 0;
        for (int t = 0; t < 256; ++t) {
            const int lane = t & 63, wave = t >> 6;
            const float v = g[t * 4];
            if (lane & 1) { if (v == 0.f) ++zeros; else if (v == -7.f) ++kept; else ++other; }
            else { if (v == 1.0f + (wave * 64 + (lane ^ 5)) * 4) ++lin; else ++other; }
        }
        printf("lds_dma_probe (%s LDS): status %s; in-range lanes lane-linear %d/128; out-of-range lanes: zeros %d, untouched %d; unexpected %d\n",
               big ? "144 KB" : "4 KB", hipGetErrorString(e), lin, zeros, kept, other);
    }
    return 0;
}
