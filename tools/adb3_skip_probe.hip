// tools/adb3_skip_probe.hip -- GPU-box probe (not part of the product library): k_conv_adirect_b3<3> on ResNet layer1's conv with one piece of the K step
// compiled out (TD_ADB3_SKIP: 1 no split, 2 no A loads, 4 no B fragment reads, 8 no weight DMA, 16 no MFMAs; results are garbage, timing only).
//   for m in 0 1 2 4 8 16 31; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTD_ADB3_SKIP=$m -Iinclude -Itdnet_amd/csrc tools/adb3_skip_probe.hip -o tools/_build/adb3_skip_$m; done
//   tools/_build/adb3_skip_<m> [H] [W] [Cin] [Cout]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "td_device.h"
#include "td_conv_ad_b3.h"

__global__ void k_fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 256, W = argc > 2 ? atoi(argv[2]) : 512, Cin = argc > 3 ? atoi(argv[3]) : 64, Cout = argc > 4 ? atoi(argv[4]) : 64;
    float *in, *out, *bias; unsigned short* wp;
    const size_t nin = (size_t)H * W * Cin, nout = (size_t)H * W * Cout, nw = conv_adb3_packed_bytes(Cout, Cin, 3) / 2;
    hipMalloc(&in, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&bias, Cout * 4); hipMalloc(&wp, nw * 2);
    k_fill<<<1024, 256>>>(in, nin, 1); hipMemset(bias, 0, Cout * 4);
    std::vector<float> w((size_t)Cout * Cin * 9);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f / 24.f - 1.f / 48.f;
    std::vector<unsigned short> packed(nw);
    conv_pack_weights_adb3(w.data(), Cout, Cin, 3, packed.data());
    hipMemcpy(wp, packed.data(), nw * 2, hipMemcpyHostToDevice);
    ConvArgs a{};
    a.in = in; a.wp = (const float*)wp; a.bias = bias; a.resid = nullptr; a.out = out; a.H = H; a.W = W; a.Cin = Cin; a.Wo = W; a.Cout = Cout; a.CoutPad = (Cout + 63) / 64 * 64;
    a.stride = 1; a.dil = 1; a.pad = 1; a.M = H * W; a.nsteps = conv_nsteps(Cin, 3, 0); a.act = 1; a.nbatch = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) conv_launch_adirect_b3(a, 3, 0, 0);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) conv_launch_adirect_b3(a, 3, 0, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("skip %2d | k_conv_adirect_b3<3> %dx%d %d->%d: %.4f ms per conv\n", TD_ADB3_SKIP, H, W, Cin, Cout, ms / 20);
    return 0;
}
