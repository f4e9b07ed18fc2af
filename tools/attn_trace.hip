// GPU-box probe: where do the waves of k_attention<1,4,4,2> (td_attn.h, the final propagation step: Lq = 32768, Lk = 2048, d_v = 512)
// spend a key super-tile?  s_memtime stamps at the loop top, after P is written, after the next tile's scores, after the barrier and
// after the P V' MFMAs; workgroups 0..7 (one CU holds two of them), the first 12 super-tiles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value tools/attn_trace.hip -o tools/_build/attn_trace
#include <hip/hip_runtime.h>
__device__ unsigned long long TD_ATTN_TRACE[8 * 4 * 12 * 5];
#define TD_ATTN_TRACE TD_ATTN_TRACE
#include "../tdnet_amd/csrc/td_device.h"
#include "../tdnet_amd/csrc/td_attn.h"
#include <cstdio>
#include <vector>
int main() {
    const int Lq = 32768, Lk = 2048, DV = 512;
    std::vector<float> q((size_t)Lq * 64), k((size_t)Lk * 64), v((size_t)attn_vp_rows(Lk) * DV), r((size_t)Lq * DV);
    unsigned st = 7u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& x : q) x = rnd(); for (auto& x : k) x = rnd(); for (auto& x : v) x = rnd(); for (auto& x : r) x = rnd();
    float *dq, *dk, *dv, *dr, *dout, *db;
    hipMalloc(&dq, q.size() * 4); hipMalloc(&dk, k.size() * 4); hipMalloc(&dv, v.size() * 4); hipMalloc(&dr, r.size() * 4); hipMalloc(&dout, r.size() * 4); hipMalloc(&db, DV * 4);
    hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dk, k.data(), k.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, r.data(), r.size() * 4, hipMemcpyHostToDevice); hipMemset(db, 0, DV * 4);
    AttnArgs a;
    a.q = dq; a.k = dk; a.vp = dv; a.bias = db; a.resid = dr; a.out = dout; a.Lq = Lq; a.Lk = Lk; a.scale_log2e = 1.4426950408889634f / 8.0f;
    a.ln_part = nullptr; a.ln_nstr = 0; a.ldv = DV;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) attn_launch(a, DV, 2, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 4; ++i) attn_launch(a, DV, 2, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("k_attention<1,4,4,2> Lq %d Lk %d: %.1f us per launch (with the stamps), %.1f TFLOP/s algorithmic\n", Lq, Lk, ms / 4 * 1e3, 2.0 * Lq * (double)Lk * (64 + DV) / (ms / 4 * 1e-3) / 1e12);
    std::vector<unsigned long long> t(8 * 4 * 12 * 5);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(TD_ATTN_TRACE), t.size() * 8);
    for (int wg = 0; wg < 8; wg += 1) {
        printf("workgroup %d: per wave, mean over super-tiles 2..11 of [A: softmax + P write, B: next scores (32 MFMAs) + maxima, barrier, P V' (256 MFMAs)] in shader cycles; period; start of tile 2\n", wg);
        for (int wv = 0; wv < 4; ++wv) {
            double d[4] = {0, 0, 0, 0}, per = 0;
            for (int s = 2; s < 12; ++s) {
                const unsigned long long* x = &t[(((size_t)wg * 4 + wv) * 12 + s) * 5];
                d[0] += (double)(x[1] - x[0]); d[1] += (double)(x[2] - x[1]); d[2] += (double)(x[3] - x[2]); d[3] += (double)(x[4] - x[3]);
                if (s > 2) per += (double)(x[0] - t[(((size_t)wg * 4 + wv) * 12 + s - 1) * 5]);
            }
            printf("  wave %d: A %6.0f  B %6.0f  barrier %6.0f  PV %6.0f   period %6.0f   t0 %llu\n", wv, d[0] / 10, d[1] / 10, d[2] / 10, d[3] / 10, per / 9,
                   t[(((size_t)wg * 4 + wv) * 12 + 2) * 5] - t[2 * 5]);
        }
    }
    return 0;
}
