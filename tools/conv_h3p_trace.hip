// GPU-box probe: where do the waves of k_conv_dma_h3p (td_conv_hd.h: four loader waves + matrix waves) spend a K step?  The kernel compiled
// with TD_P_TRACE stamps s_memtime (shader cycles) per wave and step; s_memrealtime around the launch gives the clock the chip held.
//   matrix waves: step start | first k-group's MFMAs issued | all MFMAs issued | after the barrier
//   loader waves: step start | pieces issued | after the counted vmcnt wait | after the barrier
// args: code (17 = 128 rows / 8 matrix waves, 18 = 192 rows / 6, 19 = 256 rows / 8) H W Cin Cout dil
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value tools/conv_h3p_trace.hip -o tools/_build/conv_h3p_trace
#include <hip/hip_runtime.h>
__device__ unsigned long long TD_P_TRACE[4 * 16 * 24 * 4];
#define TD_P_TRACE TD_P_TRACE
#include "../tdnet_amd/csrc/td_device.h"
#include "../tdnet_amd/csrc/td_conv_hd.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    const int code = argc > 1 ? atoi(argv[1]) : CD_128_P;
    const int H = argc > 2 ? atoi(argv[2]) : 90, W = argc > 3 ? atoi(argv[3]) : 120, Cin = argc > 4 ? atoi(argv[4]) : 256,
              Cout = argc > 5 ? atoi(argv[5]) : 256, KS = 3, dil = argc > 6 ? atoi(argv[6]) : 2;
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
    auto launch = [&]() { return conv_launch_dma3p(a, code, KS, true, 0); };
    for (int i = 0; i < 3; ++i) if (!launch()) { printf("shape does not run on the loader-wave kernel\n"); return 1; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("tile code %d, %d x %d x %d -> %d, dilation %d, %d K steps: %.1f us per launch (with the stamps), %.0f TFLOP/s\n", code, H, W, Cin, Cout, dil, nsteps, ms / 5 * 1e3,
           2.0 * H * W * Cin * 9.0 * Cout / (ms / 5 * 1e-3) / 1e12);
    std::vector<unsigned long long> t(4 * 16 * 24 * 4);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(TD_P_TRACE), t.size() * 8);
    const int nwc = code == CD_192_P ? 6 : 8, nld = 4;
    const int ns = nsteps < 24 ? nsteps : 24;
    for (int wg = 0; wg < 2; ++wg) {
        printf("workgroup %d: mean over steps 3..%d, shader cycles.  matrix waves: [to first group issued, rest of the MFMAs issued, to the barrier's end]; loaders: [issue, vmcnt wait, barrier]\n", wg, ns - 1);
        for (int wv = 0; wv < nwc + nld; ++wv) {
            double d[3] = {0, 0, 0}, per = 0;
            int cnt = 0;
            for (int s = 3; s < ns; ++s, ++cnt) {
                const unsigned long long* q = &t[(((size_t)wg * 16 + wv) * 24 + s) * 4];
                const unsigned long long prev = t[(((size_t)wg * 16 + wv) * 24 + s - 1) * 4 + 3];
                d[0] += (double)(q[1] - q[0]); d[1] += (double)(q[2] - q[1]); d[2] += (double)(q[3] - q[2]);
                per += (double)(q[3] - prev);
            }
            printf("  %s wave %2d: %6.0f %6.0f %6.0f   period %6.0f\n", wv < nwc ? "matrix" : "loader", wv, d[0] / cnt, d[1] / cnt, d[2] / cnt, per / cnt);
        }
    }
    printf("workgroup 0, steps 6..8, stamps relative to step 6's start of matrix wave 0:\n");
    const unsigned long long base = t[(6) * 4 + 0];
    for (int wv = 0; wv < nwc + nld; ++wv) {
        printf("  wave %2d:", wv);
        for (int s = 6; s < 9; ++s) for (int k = 0; k < 4; ++k) printf(" %6lld", (long long)(t[(((size_t)0 * 16 + wv) * 24 + s) * 4 + k] - base));
        printf("\n");
    }
    return 0;
}
