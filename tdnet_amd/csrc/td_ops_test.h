// td_ops_test.h -- single-operator entry points (tests/ check each kernel family against torch fp32 through them) and the roofline /
// tuning probes of include/tdnet_test.h.  NOT in the product library: libtdnet_hip.so (td_model.hip) is built without this file; the
// tests' libtdnet_hip_test.so (td_model_test.hip = td_model.hip + this file) and the emulator library carry it.
#pragma once
#include "td_frame.h"

// ---------------------------------------------------------------------------------------------------------------
// single-operator entry points (tests)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int tdnet_op_conv2d(const float* in, int H, int W, int Cin, const float* w_host, const float* bias_host, int Cout, int KS,
                               int stride, int dil, const float* resid, int act, const tdnet_opts* opts, int tile, float* out,
                               void* stream) {
    // tile < 0: the heuristic's tile for this shape; 0..CT_COUNT-1: forced (0: 128x128, 1: 64x128, 2: 128x64, 3..5: the same on the
    // two-stage pipeline) -- lets tests cover every variant
    if (KS != 1 && KS != 3) return td_fail("tdnet_op_conv2d: KS must be 1 or 3");
    if (tile >= CT_COUNT) return td_fail("tdnet_op_conv2d: tile must be < %d", CT_COUNT);
    const tdnet_opts o = opts_or_default(opts);
    ConvLayer L;
    std::vector<float> w(w_host, w_host + (size_t)Cout * Cin * KS * KS), b;
    if (bias_host) b.assign(bias_host, bias_host + Cout);
    const int pad = dil * (KS / 2);
    const long M = (long)out_size(H, KS, stride, dil, pad) * out_size(W, KS, stride, dil, pad);
    // tdnet_opts.overlap bit 1: an even-dilation Winograd conv runs as its two row-parity chunks (here one after the other)
    if (make_conv_layer(L, w, b, Cout, Cin, KS, stride, dil, act, false, M, o, tile < 0 ? -1 : tile, !(o.overlap & 1) ? 1 : ((o.overlap & 64) && dil % 4 == 0) ? 4 : 2)) return -1;
    int rc = run_conv(nullptr, L, in, H, W, resid, out, (hipStream_t)stream);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = td_fail("tdnet_op_conv2d: device error");
    free_conv_layer(L);
    return rc;
}
// Test entry for the fp16-activation storage of tdnet_opts.precision = 1: the fp32 arguments are rounded to fp16 maps in HBM, the
// conv runs with fp16 input / residual / output (k_conv_igemm_h<.., IN16, OUT16>), and the fp16 result is widened into out.
extern "C" int tdnet_op_conv2d_f16io(const float* in, int H, int W, int Cin, const float* w_host, const float* bias_host, int Cout,
                                     int KS, int stride, int dil, const float* resid, int act, int tile, float* out, void* stream) {
    if (KS != 1 && KS != 3) return td_fail("tdnet_op_conv2d_f16io: KS must be 1 or 3");
    if (Cin % 64) return td_fail("tdnet_op_conv2d_f16io: Cin must be a multiple of 64");
    // tile 16 / 17 / 18 / 19: the LDS-DMA kernel with 128 / 192 / 256-row tiles, 256 x 256 (td_conv_hd.h); 20 / 21: 128 rows on a ring of
    // four / two LDS buffers whatever the grid (16 chooses by the grid); 22: 128 rows, eight waves; 23 / 26: the same on row images with one
    // barrier per super-step / per K step only; 24 / 25: 192 rows likewise; 27 / 28 / 29: 256 / 192 / 128 rows in the early-landing form
    // only; -1: the heuristic (DMA kernel where it applies)
    const bool no_rowimg = tile >= 48 && tile <= 61;                   // 48 + code: the same tile, tap-by-tap staging (k_conv_dma_h) instead of row images
    if (no_rowimg) tile -= 32;
    static const int code_of_tile[14] = {CD_128, CD_192, CD_256, CD_256x256, CD_128_4BUF, CD_128_2BUF, CD_128_8W,            // 16 .. 22
                                         CD_128_SUPER, CD_192_SUPER, CD_192_STEP, CD_128_STEP, CD_256_EARLY, CD_192_EARLY, CD_128_EARLY};   // 23 .. 29
    static const int code_of_tile_p[6] = {CD_128_P, CD_192_P, CD_256_P,        // 31 .. 33: row images with four dedicated loader waves (k_conv_dma_h3p)
                                          CD_128_N, CD_192_N, CD_256_N};       // 34 .. 36: narrow tiles, rows x 64 channels (k_conv_dma_h3n)
    if (tile == 30) return td_fail("tdnet_op_conv2d_f16io: tile 30 (the weights-resident 64 -> 64 kernel) was removed in round 5");
    const int force_rh = tile >= 16 && tile <= 29 ? code_of_tile[tile - 16] : tile >= 31 && tile <= 36 ? code_of_tile_p[tile - 31] : 0;
    if (force_rh) tile = CT_128x128_DEEP;
    if (tile >= CT_COUNT) return td_fail("tdnet_op_conv2d_f16io: tile must be < %d or 16..36 (+ 32 for 16..29)", CT_COUNT);
    hipStream_t s = (hipStream_t)stream;
    tdnet_opts o = opts_or_default(nullptr);
    o.precision = 1;
    ConvLayer L;
    std::vector<float> w(w_host, w_host + (size_t)Cout * Cin * KS * KS), b;
    if (bias_host) b.assign(bias_host, bias_host + Cout);
    const int pad = dil * (KS / 2);
    const int Ho = out_size(H, KS, stride, dil, pad), Wo = out_size(W, KS, stride, dil, pad);
    if (make_conv_layer(L, w, b, Cout, Cin, KS, stride, dil, act, false, (long)Ho * Wo, o, tile < 0 ? -1 : tile)) return -1;
    L.in16 = L.out16 = true;
    L.rowimg_off = no_rowimg;
    if (force_rh == CD_256x256 && L.CoutPad % 256) { free_conv_layer(L); return td_fail("tdnet_op_conv2d_f16io: the 256 x 256 tile needs Cout padded to a multiple of 256"); }
    if (force_rh && !conv_dma_supports(Cin, Cout, KS, L.tile)) {
        free_conv_layer(L);
        return td_fail("tdnet_op_conv2d_f16io: this shape cannot run on the LDS-DMA kernel");
    }
    if (force_rh) L.rh = force_rh;
    else if (tile < 0 && Cout >= 128 && conv_dma_supports(Cin, Cout, KS, L.tile)) L.rh = conv_dma_pick_rh((long)Ho * Wo, Cout, L.CoutPad % 256 == 0);
    _Float16 *hin = nullptr, *hres = nullptr, *hout = nullptr;
    const long nin = (long)H * W * Cin, nout = (long)Ho * Wo * Cout;
    auto cleanup = [&]() {                                             // one release path, also for the error returns
        for (_Float16* q : {hin, hout, hres}) if (q) hipFree(q);
        free_conv_layer(L);
    };
    if (dev_alloc(&hin, (size_t)nin) || dev_alloc(&hout, (size_t)nout) || (resid && dev_alloc(&hres, (size_t)nout))) { cleanup(); return -1; }
    TD_LAUNCH(k_f2h, dim3(td_grid_for(nin)), dim3(256), 0, s, in, hin, nin);
    if (resid) TD_LAUNCH(k_f2h, dim3(td_grid_for(nout)), dim3(256), 0, s, resid, hres, nout);
    int rc = run_conv(nullptr, L, (const float*)hin, H, W, (const float*)hres, (float*)hout, s);
    TD_LAUNCH(k_h2f, dim3(td_grid_for(nout)), dim3(256), 0, s, (const _Float16*)hout, out, nout);
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) rc = td_fail("tdnet_op_conv2d_f16io: device error");
    cleanup();
    return rc;
}
extern "C" int tdnet_op_stem(const float* img, int H, int W, const float* w_host, const float* bias_host, const tdnet_opts* opts,
                             float* out, void* stream) {
    const tdnet_opts o = opts_or_default(opts);
    hipStream_t s = (hipStream_t)stream;
    const int H1 = (H - 1) / 2 + 1, W1 = (W - 1) / 2 + 1;
    ConvLayer L;
    std::vector<float> w(w_host, w_host + 64 * 3 * 49), b;
    if (bias_host) b.assign(bias_host, bias_host + 64);
    if (make_conv_layer(L, w, b, 64, 3, 7, 2, 1, 1, true, (long)H1 * W1, o)) return -1;
    float *img4 = nullptr, *s1 = nullptr;
    const size_t img_floats = std::max((size_t)H * W * 4, (size_t)stem_rows_hp(H) * stem_rows_wp(W) * 3 + 4);
    if (dev_alloc(&img4, img_floats) || dev_alloc(&s1, (size_t)H1 * W1 * 64)) return -1;
    if (L.stem_rows) TD_HIP(hipMemsetAsync(img4, 0, img_floats * sizeof(float), s));   // the packed-row image's zero border
    run_stem_pre(nullptr, img, H, W, img4, s, o.fusion, L.stem_rows);
    run_conv(nullptr, L, img4, H, W, nullptr, s1, s);
    run_maxpool(nullptr, s1, H1, W1, 64, out, s, o.fusion);
    TD_HIP(hipStreamSynchronize(s));
    TD_HIP(hipGetLastError());
    hipFree(img4); hipFree(s1);
    free_conv_layer(L);
    return 0;
}
extern "C" int tdnet_op_attention(const float* q, const float* k, const float* vp, const float* bias, const float* resid, int Lq,
                                  int Lk, int DV, int online, const float* ln_g, const float* ln_b, float* ln_out, float* out,
                                  void* stream) {
    if (Lk < 1 || Lq < 1) return td_fail("tdnet_op_attention: empty input");
    hipStream_t s = (hipStream_t)stream;
    const bool padded = (online & 32) != 0;                            // online | 32: the caller's vp already has the padding rows (probes that time the kernel)
    const bool slices = (online & 64) != 0;                            // online | 64: DV = 512 as two 256-channel slices in one launch (the chain's cached-frame steps)
    online &= ~(32 | 64);
    // arguments are validated BEFORE anything is allocated; every later exit goes through cleanup()
    if (ln_out && (!ln_g || !ln_b)) return td_fail("tdnet_op_attention: ln_out needs ln_g and ln_b");
    const bool split = online >= 17 && online <= 19;                   // 17: the split kernels of precision 2 (td_attn_b3.h), form by size; 18: the 64-query form; 19: the 32-query form
    if (online != 16 && !split && (online < 0 || online > 2)) return td_fail("tdnet_op_attention: online must be 0, 1, 2 or 16 .. 19");
    float *part = nullptr, *mean = nullptr, *rstd = nullptr, *vpad = nullptr;
    _Float16* vt = nullptr;                                            // online == 16: the fp16-MFMA kernel of tdnet_opts.precision = 1 (td_attn_h.h)
    auto cleanup = [&]() {
        for (float* q2 : {part, mean, rstd, vpad}) if (q2) hipFree(q2);
        if (vt) hipFree(vt);
    };
    int rc = 0;
    if (ln_out && (dev_alloc(&part, (size_t)2 * attn_strips(Lq, DV) * DV) || dev_alloc(&mean, DV) || dev_alloc(&rstd, DV))) rc = -1;   // + plane LayerNorm of the result from the epilogue's strip statistics
    if (!rc && (online == 16 || split) && dev_alloc(&vt, (size_t)(split ? 3 : 1) * DV * attn_lkpad(Lk))) rc = -1;
    if (!rc && online != 16 && !split && !padded && attn_vp_rows(Lk) != Lk) {    // the kernels' contract: V' padded to attn_vp_rows(Lk) zero rows
        const size_t rows = (size_t)attn_vp_rows(Lk);
        if (dev_alloc(&vpad, rows * DV)) rc = -1;
        else if (hipMemsetAsync(vpad, 0, rows * DV * sizeof(float), s) != hipSuccess ||
                 hipMemcpyAsync(vpad, vp, (size_t)Lk * DV * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = td_fail("tdnet_op_attention: copy failed");
        else vp = vpad;
    }
    if (!rc) rc = run_attention(nullptr, q, k, vp, bias, resid, Lq, Lk, DV, out, s, online >= 16 ? 1 : online, part, vt, slices, false, online == 17 ? 1 : online == 18 ? 3 : online == 19 ? 2 : 0);
    if (!rc && ln_out) run_layernorm(nullptr, out, Lq, DV, ln_g, ln_b, part, mean, rstd, ln_out, s, attn_strips(Lq, DV));
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) rc = td_fail("tdnet_op_attention: device error");
    cleanup();
    return rc;
}
extern "C" int tdnet_op_layernorm_hw(const float* x, int HW, int C, const float* g, const float* b, float* out, void* stream) {
    if (C % 4 || (C / 4 <= 256 ? 256 % (C / 4) != 0 : C / 4 > 512)) return td_fail("tdnet_op_layernorm_hw: C must be one of 4*{1,2,4,...,256} or 2048");
    float *part = nullptr, *mean = nullptr, *rstd = nullptr;
    if (dev_alloc(&part, (size_t)2 * 512 * C) || dev_alloc(&mean, C) || dev_alloc(&rstd, C)) return -1;
    run_layernorm(nullptr, x, HW, C, g, b, part, mean, rstd, out, (hipStream_t)stream);
    TD_HIP(hipStreamSynchronize((hipStream_t)stream));
    TD_HIP(hipGetLastError());
    hipFree(part); hipFree(mean); hipFree(rstd);
    return 0;
}
extern "C" int tdnet_op_ppm(const float* c4, int h, int w, const float* w_host, const float* b_host, int path_num, int pid, float* z,
                            void* stream) {
    const int C = 512, FS = C / (path_num * 4);
    if (path_num != 2) return td_fail("tdnet_op_ppm: path_num must be 2 (td4 passes path_num//2, td2 passes 2)");
    std::vector<float> pw((size_t)4 * FS * C), pb((size_t)4 * FS);
    for (int j = 0; j < 4; ++j)
        for (int o = 0; o < FS; ++o) {
            for (int c = 0; c < C; ++c) pw[((size_t)j * C + c) * FS + o] = w_host[((size_t)j * 128 + pid * FS + o) * C + c];
            pb[j * FS + o] = b_host[j * 128 + pid * FS + o];
        }
    float *dw = nullptr, *db = nullptr, *rowpart = nullptr, *pooled = nullptr, *ppmfeat = nullptr;
    if (upload(&dw, pw) || upload(&db, pb)) return -1;
    if (dev_alloc(&rowpart, (size_t)h * 36 * C) || dev_alloc(&pooled, 50 * C) || dev_alloc(&ppmfeat, 50 * FS)) return -1;
    run_ppm(nullptr, c4, h, w, C, C / 2, FS, dw, db, pid, rowpart, pooled, ppmfeat, z, (hipStream_t)stream);
    TD_HIP(hipStreamSynchronize((hipStream_t)stream));
    TD_HIP(hipGetLastError());
    for (float* q : {dw, db, rowpart, pooled, ppmfeat}) hipFree(q);
    return 0;
}
extern "C" int tdnet_op_upsample(const float* in, int C, int h, int w, int H, int W, float* out, void* stream) {
    launch_upsample(in, C, h, w, H, W, out, (hipStream_t)stream);
    TD_HIP(hipStreamSynchronize((hipStream_t)stream));
    TD_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// tuning / roofline hooks (not on the product path)
// ---------------------------------------------------------------------------------------------------------------
// Pure-MFMA loop: the practical fp32-MFMA ceiling of THIS chip at its sustained clock (4 independent accumulators per
// wave, `waves_per_simd` waves per SIMD, no memory traffic).
TD_KERNEL void k_mfma_peak(float* out, int iters) {
    f32x16 a0, a1, a2, a3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 1.f; a2[r] = 2.f; a3[r] = 3.f; }
    float x = 1.0f + (float)(threadIdx.x & 7) * 1e-3f, y = 1.0f - (float)(threadIdx.x & 3) * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = td_mfma32(x, y, a0); a1 = td_mfma32(y, x, a1); a2 = td_mfma32(x, x, a2); a3 = td_mfma32(y, y, a3);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[0] = s;                      // keep the accumulators live
}
// returns achieved TFLOP/s (fp32 MFMA) or <0
extern "C" double tdnet_bench_mfma_peak(int waves_per_simd, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* d = nullptr;
    if (hipMalloc((void**)&d, 256) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;            // 256 CUs x (4 SIMDs = one 256-thread block) x waves_per_simd
    TD_LAUNCH(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, 16);
    hipEventRecord(e0, s);
    TD_LAUNCH(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, iters);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    const double flop = (double)blocks * 4 /*waves*/ * (double)iters * 32 /*mfma per iter*/ * 4096.0;
    return ms > 0.f ? flop / (ms * 1e-3) / 1e12 : -1.0;
}
// Average device time (ms, HIP events on `stream`) of `iters` launches of one conv configuration on random data.
extern "C" double tdnet_bench_conv(int H, int W, int Cin, int Cout, int KS, int stride, int dil, int tile, int iters,
                                   const tdnet_opts* opts, void* stream) {
    const tdnet_opts o = opts_or_default(opts);
    if ((KS != 1 && KS != 3) || Cin % 32 || tile < -1 || tile >= CT_COUNT) { td_fail("tdnet_bench_conv: bad arguments"); return -1.0; }
    hipStream_t s = (hipStream_t)stream;
    ConvLayer L;
    std::vector<float> w((size_t)Cout * Cin * KS * KS), x((size_t)H * W * Cin), b(Cout, 0.1f);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : w) v = rnd() * 0.05f;
    for (auto& v : x) v = rnd();
    const int pad_ = dil * (KS / 2);
    const long M_ = (long)out_size(H, KS, stride, dil, pad_) * out_size(W, KS, stride, dil, pad_);
    // tile -1: the heuristic's choice for this M; overlap bit 1 (+ 64): the conv as its 2 (4) row classes one after the other on the stream
    if (make_conv_layer(L, w, b, Cout, Cin, KS, stride, dil, 1, false, M_, o, tile, !(o.overlap & 1) ? 1 : ((o.overlap & 64) && dil % 4 == 0) ? 4 : 2)) return -1.0;
    float *din = nullptr, *dout = nullptr;
    if (upload(&din, x)) return -1.0;
    const int Ho = out_size(H, KS, stride, dil, L.pad), Wo = out_size(W, KS, stride, dil, L.pad);
    if (dev_alloc(&dout, (size_t)Ho * Wo * Cout)) return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    TdWeights none;
    tdnet tmp(&none);                                                 // only carries the Winograd workspace for run_conv
    if (L.wino) {
        const size_t T = (size_t)wino_tiles(H, W, dil, L.wino), nb = (size_t)(L.wino + 2) * (L.wino + 2);
        tmp.wino_v_floats = nb * (T + L.wino_pad) * Cin; tmp.wino_m_floats = nb * (T + L.wino_pad) * Cout;
        if (dev_alloc(&tmp.wino_v, tmp.wino_v_floats) || dev_alloc(&tmp.wino_m, tmp.wino_m_floats)) return -1.0;
    }
    tdnet* ws = L.wino ? &tmp : nullptr;
    for (int i = 0; i < 2; ++i) run_conv(ws, L, din, H, W, nullptr, dout, s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) run_conv(ws, L, din, H, W, nullptr, dout, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(din); hipFree(dout);
    if (tmp.wino_v) hipFree(tmp.wino_v);
    if (tmp.wino_m) hipFree(tmp.wino_m);
    free_conv_layer(L);
    return ms / iters;
}
