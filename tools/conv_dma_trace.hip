// GPU-box probe: where do the waves of k_conv_dma_h (td_conv_hd.h) spend a K step?  The kernel compiled with TD_DMA_TRACE stamps
// s_memtime after the DMA issue, after the MFMAs, after the vmcnt wait and after the barrier, for workgroups 0..3, every wave, the
// first 24 steps, on the dominant layer4 shape (128x256x512 -> 512, 3x3, dilation 4, fp16 maps) or the one given: code H W Cin Cout dil.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value tools/conv_dma_trace.hip -o tools/_build/conv_dma_trace
#include <hip/hip_runtime.h>
__device__ unsigned long long TD_DMA_TRACE[4 * 8 * 24 * 4];
#define TD_DMA_TRACE TD_DMA_TRACE
#include "../tdnet_amd/csrc/td_device.h"
#include "../tdnet_amd/csrc/td_conv_hd.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    const int code = argc > 1 ? atoi(argv[1]) : 8;                    // 8 = 256 x 256, 4 = 256 x 128, 3, 2 (5 / 6: two / four buffers), 7 = 128 x 128 with eight waves
    const int H = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 256, Cin = argc > 4 ? atoi(argv[4]) : 512,
              Cout = argc > 5 ? atoi(argv[5]) : 512, KS = 3, dil = argc > 6 ? atoi(argv[6]) : 4;
    std::vector<float> w((size_t)Cout * Cin * 9);
    unsigned st = 1u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : w) v = rnd() * 0.02f;
    const ConvTile tile = CT_128x128_DEEP;
    const int CoutPad = conv_cout_pad(Cout, tile), nsteps = conv_nsteps_h(Cin, KS);
    std::vector<_Float16> packed((size_t)nsteps * 8 * CoutPad * 8), x((size_t)H * W * Cin);
    conv_pack_weights_h(w.data(), Cout, Cin, KS, tile, packed.data());
    for (auto& v : x) v = (_Float16)rnd();
    _Float16 *dx, *dw, *dout; float* db;
    hipMalloc(&dx, x.size() * 2); hipMalloc(&dw, packed.size() * 2); hipMalloc(&dout, (size_t)H * W * Cout * 2); hipMalloc(&db, Cout * 4);
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
    hipMemset(db, 0, Cout * 4);
    ConvArgs a;
    a.in = (const float*)dx; a.wp = (const float*)dw; a.bias = db; a.resid = nullptr; a.out = (float*)dout;
    a.H = H; a.W = W; a.Cin = Cin; a.Wo = W; a.Cout = Cout; a.CoutPad = CoutPad; a.stride = 1; a.dil = dil; a.pad = dil; a.M = H * W;
    a.nsteps = nsteps; a.act = 1; a.tiles_n = 0; a.nbatch = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) conv_launch_dma(a, code, KS, true, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) conv_launch_dma(a, code, KS, true, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("tile code %d: %.1f us per launch (with the stamps), %.0f TFLOP/s\n", code, ms / 5 * 1e3, 2.0 * H * W * Cin * 9.0 * Cout / (ms / 5 * 1e-3) / 1e12);
    std::vector<unsigned long long> t(4 * 8 * 24 * 4);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(TD_DMA_TRACE), t.size() * 8);
    const int nw = code == 8 || code == 4 || code == 7 ? 8 : code >= 5 ? 4 : 2 * code;
    printf("shape %d x %d x %d -> %d, dilation %d: %d K steps per tile\n", H, W, Cin, Cout, dil, nsteps);
    for (int wg = 0; wg < 2; ++wg) {
        printf("workgroup %d: per wave, mean over steps 4..23 of [issue, MFMAs, vmcnt wait, barrier] in shader cycles, and the step period\n", wg);
        for (int wv = 0; wv < nw; ++wv) {
            double d[4] = {0, 0, 0, 0}, per = 0;
            for (int s = 4; s < 24; ++s) {
                const unsigned long long* q = &t[(((size_t)wg * 8 + wv) * 24 + s) * 4];
                const unsigned long long prev = t[(((size_t)wg * 8 + wv) * 24 + s - 1) * 4 + 3];
                d[0] += (double)(q[0] - prev); d[1] += (double)(q[1] - q[0]); d[2] += (double)(q[2] - q[1]); d[3] += (double)(q[3] - q[2]);
                per += (double)(q[3] - prev);
            }
            printf("  wave %d: issue %6.0f  mfma %6.0f  vmwait %6.0f  barrier %6.0f   period %6.0f\n", wv, d[0] / 20, d[1] / 20, d[2] / 20, d[3] / 20, per / 20);
        }
    }
    const unsigned long long* q0 = &t[0];
    printf("workgroup 0 wave 0 raw stamps of steps 8..11 relative to step 8's start:");
    const unsigned long long base = t[(7) * 4 + 3];
    for (int s = 8; s < 12; ++s) for (int k = 0; k < 4; ++k) printf(" %llu", t[s * 4 + k] - base);
    printf("\n");
    (void)q0;
    return 0;
}
