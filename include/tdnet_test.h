/*
 * tdnet_test.h -- entry points of libtdnet_hip_test.so ONLY: single-operator calls (tests/ check each kernel family against torch fp32 / fp64
 * through them) and tuning probes.  libtdnet_hip_test.so is built from the same sources as the product library plus tdnet_amd/csrc/td_ops_test.h
 * and exports everything include/tdnet.h declares as well; the PRODUCT library libtdnet_hip.so exports none of the names below.
 */
#ifndef TDNET_TEST_H
#define TDNET_TEST_H
#include "tdnet.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Roofline / tuning probes: sustained fp32-MFMA TFLOP/s of a register-only MFMA loop, and the average device ms of
 * `iters` launches of one conv configuration (random data, tile as in tdnet_op_conv2d).                             */
double tdnet_bench_mfma_peak(int waves_per_simd, int iters, void* stream);
double tdnet_bench_conv(int H, int W, int Cin, int Cout, int KS, int stride, int dil, int tile /* -1: heuristic */, int iters,
                        const tdnet_opts* opts /* NULL = defaults */, void* stream);

/* ---- single-operator entry points (used by tests/ to check each kernel family against torch fp32) -------------- */
/* NHWC conv: in [H,W,Cin] dev, weight OIHW host [Cout,Cin,KS,KS], bias host [Cout] or NULL, residual dev
 * [Ho,Wo,Cout] or NULL, act 0 none / 1 ReLU / 2 LeakyReLU(0.01); out [Ho,Wo,Cout] dev.                           */
int tdnet_op_conv2d(const float* in_dev, int H, int W, int Cin, const float* w_host, const float* bias_host,
                    int Cout, int KS, int stride, int dil, const float* resid_dev, int act,
                    const tdnet_opts* opts /* NULL = defaults */,
                    int tile /* -1 = heuristic; 0: 128x128, 1: 64x128, 2: 128x64, 3..5: the same on the two-stage pipeline */,
                    float* out_dev, void* stream);
/* the same conv with the fp16 activation STORAGE of tdnet_opts.precision = 1: in / resid are rounded to fp16 maps in HBM, the kernel
 * reads and writes fp16 (fp16 MFMA, fp32 accumulate), the fp16 result is widened into out_dev.  Cin % 64 == 0.                    */
int tdnet_op_conv2d_f16io(const float* in_dev, int H, int W, int Cin, const float* w_host, const float* bias_host,
                          int Cout, int KS, int stride, int dil, const float* resid_dev, int act, int tile, float* out_dev, void* stream);
/* stem: NCHW image [3,H,W] -> conv7x7 s2 p3 (+bias) -> ReLU -> maxpool3x3 s2 p1 -> NHWC [H2,W2,64] (resnet.py:205-208) */
int tdnet_op_stem(const float* img_dev, int H, int W, const float* w_host, const float* bias_host,
                  const tdnet_opts* opts /* NULL = defaults */, float* out_dev, void* stream);
/* softmax(q k^T / sqrt(dk)) v' + bias + resid: q [Lq,64], k [Lk,64], vp [Lk,DV], bias dev [DV]|NULL, resid [Lq,DV]|NULL.
 * online: tdnet_opts.attention (0 = exact two-pass softmax, 1 = single pass with a lazily moved reference, 2 = 1 pipelined to one
 * barrier per key tile), or 16 = the fp16-MFMA kernel of tdnet_opts.precision = 1 (single pass; operands and P rounded to fp16,
 * softmax and accumulation fp32), or 17 = the split kernel of tdnet_opts.precision = 2 (td_attn_b3.h: q, k, P and v' as three bf16 parts each, six
 * bf16-MFMA products per product, fp32 softmax and accumulation; at DV = 512 k the form is picked by Lq as in a frame; 18 forces the 64-query / eight-wave
 * form, 19 the 32-query form).  The fp32 kernels read V' in whole 128-key tiles: when Lk is not a multiple of 128 the op runs on a
 * zero-padded copy of vp, unless `online | 32` says vp_dev itself has ((Lk + 127) / 128) * 128 rows, the extra ones finite.
 * ln_out != NULL: also the plane LayerNorm (affine ln_g, ln_b [Lq]) of the result, from the strip statistics the kernel's epilogue
 * writes (tdnet_opts.fusion bit 2) -> ln_out [Lq,DV].                                                                          */
int tdnet_op_attention(const float* q_dev, const float* k_dev, const float* vp_dev, const float* bias_dev,
                       const float* resid_dev, int Lq, int Lk, int DV, int online, const float* ln_g_dev, const float* ln_b_dev,
                       float* ln_out_dev, float* out_dev, void* stream);
/* LayerNorm over the (h,w) plane of every channel, affine g,b [h*w] shared by channels (td4_psp18.py:306-312); NHWC */
int tdnet_op_layernorm_hw(const float* x_dev, int HW, int C, const float* g_dev, const float* b_dev, float* out_dev, void* stream);
/* PPM (td4_psp18.py:271-284): c4 NHWC [h,w,512] -> z NHWC [h,w,512]; w_host: 4 folded [128,512] matrices, b_host 4x[128] */
int tdnet_op_ppm(const float* c4_dev, int h, int w, const float* w_host, const float* b_host, int path_num, int pid,
                 float* z_dev, void* stream);
/* bilinear align_corners=True (td4_psp18.py:227): planar [C,h,w] -> [C,H,W]                                       */
int tdnet_op_upsample(const float* in_dev, int C, int h, int w, int H, int W, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
