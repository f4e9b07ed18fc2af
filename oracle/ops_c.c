/* oracle/ops_c.c -- CPU ORACLE (test infrastructure, NOT product code): plain-C restatement of the L0 operators the
 * reference's hot path calls in PyTorch.  The reference's arithmetic lives in a third-party dependency that is not under
 * /root/reference (PyTorch: README.md:17 "Pytorch 1.1.0", requirements.txt); this file restates the PUBLISHED
 * semantics of the operators at the reference's call sites, so that oracle/tdnet_ref.py can run its graph without any PyTorch
 * kernel (oracle/c_ops.py binds it; tdnet_ref.set_ops(COps)).  Plain loops, fp32 in/out, fp64 accumulation; nothing here is
 * shared with the HIP kernels.  Only tests/ build and call it (tests/test_oracle_c.py pins it against the golden vectors
 * captured from the real reference and against PyTorch's operators).
 *
 * Call sites restated (paths relative to /root/reference/Testing/model/pspnet):
 *   tdc_conv2d              nn.Conv2d                  resnet.py:32-37,127-136,172-177; transformer.py:153-155; td4_psp18.py:255-262,295-299
 *   tdc_max_pool2d          nn.MaxPool2d(3,2,1)        resnet.py:137        (padding acts as -inf, floor mode)
 *   tdc_adaptive_avg_pool2d nn.AdaptiveAvgPool2d(o)    td4_psp18.py:250-253 (bin i = [floor(i n/o), ceil((i+1) n/o)))
 *   tdc_bilinear_ac         F.interpolate(bilinear, align_corners=True)     td4_psp18.py:27,227,273-276
 *   tdc_bmm_nn / tdc_bmm_nt torch.bmm                  transformer.py:132,137
 *   tdc_softmax_lastdim     nn.Softmax(dim=2)          transformer.py:124,134
 *   tdc_layer_norm_plane    nn.LayerNorm([h,w])        td4_psp18.py:306-312 (biased variance, eps inside the sqrt)
 * All tensors are dense row-major; images are NCHW with N = 1 folded into C by the caller.
 */
#include <math.h>
#include <stddef.h>

/* y[o][oy][ox] = b[o] + sum_{c,ky,kx} w[o][c][ky][kx] * x[c][oy*s - p + ky*d][ox*s - p + kx*d]   (zero padding) */
void tdc_conv2d(const float* x, int C, int H, int W, const float* w, const float* b, int O, int KS, int stride, int pad, int dil,
                float* y, int Ho, int Wo) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int o = 0; o < O; ++o)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                double acc = b ? (double)b[o] : 0.0;
                for (int c = 0; c < C; ++c)
                    for (int ky = 0; ky < KS; ++ky) {
                        const int iy = oy * stride - pad + ky * dil;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < KS; ++kx) {
                            const int ix = ox * stride - pad + kx * dil;
                            if (ix < 0 || ix >= W) continue;
                            acc += (double)w[(((size_t)o * C + c) * KS + ky) * KS + kx] * (double)x[((size_t)c * H + iy) * W + ix];
                        }
                    }
                y[((size_t)o * Ho + oy) * Wo + ox] = (float)acc;
            }
}

/* max over the k x k window, positions outside the image do not take part (-inf padding), floor output size */
void tdc_max_pool2d(const float* x, int C, int H, int W, int k, int stride, int pad, float* y, int Ho, int Wo) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float m = -INFINITY;
                for (int ky = 0; ky < k; ++ky) {
                    const int iy = oy * stride - pad + ky;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        const int ix = ox * stride - pad + kx;
                        if (ix < 0 || ix >= W) continue;
                        const float v = x[((size_t)c * H + iy) * W + ix];
                        if (v > m) m = v;
                    }
                }
                y[((size_t)c * Ho + oy) * Wo + ox] = m;
            }
}

/* bin i of an axis of length n covers [floor(i n / o), ceil((i+1) n / o)); plain mean; bins may overlap */
void tdc_adaptive_avg_pool2d(const float* x, int C, int H, int W, int o, float* y) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int by = 0; by < o; ++by) {
            const int y0 = (by * H) / o, y1 = ((by + 1) * H + o - 1) / o;
            for (int bx = 0; bx < o; ++bx) {
                const int x0 = (bx * W) / o, x1 = ((bx + 1) * W + o - 1) / o;
                double s = 0.0;
                for (int iy = y0; iy < y1; ++iy)
                    for (int ix = x0; ix < x1; ++ix) s += (double)x[((size_t)c * H + iy) * W + ix];
                y[((size_t)c * o + by) * o + bx] = (float)(s / (double)((y1 - y0) * (x1 - x0)));
            }
        }
}

/* align_corners=True: src = dst (n_in - 1)/(n_out - 1) (0 when n_out == 1); i0 = floor(src), i1 = min(i0+1, n_in-1); separable.
 * The coordinate arithmetic is float32 like PyTorch's (area_pixel_compute_source_index), the blend is evaluated in fp64. */
void tdc_bilinear_ac(const float* x, int C, int h, int w, float* y, int H, int W) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int Y = 0; Y < H; ++Y) {
            const float fy = sy * (float)Y;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const double ly = (double)(fy - (float)y0);
            for (int X = 0; X < W; ++X) {
                const float fx = sx * (float)X;
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const double lx = (double)(fx - (float)x0);
                const float* p = x + (size_t)c * h * w;
                const double top = (1.0 - lx) * (double)p[(size_t)y0 * w + x0] + lx * (double)p[(size_t)y0 * w + x1];
                const double bot = (1.0 - lx) * (double)p[(size_t)y1 * w + x0] + lx * (double)p[(size_t)y1 * w + x1];
                y[((size_t)c * H + Y) * W + X] = (float)((1.0 - ly) * top + ly * bot);
            }
        }
}

/* c[m][n] = sum_k a[m][k] b[k][n] */
void tdc_bmm_nn(const float* a, const float* b, float* c, int M, int K, int N) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += (double)a[(size_t)m * K + k] * (double)b[(size_t)k * N + n];
            c[(size_t)m * N + n] = (float)s;
        }
}
/* c[m][n] = sum_k a[m][k] b[n][k]   (q k^T) */
void tdc_bmm_nt(const float* a, const float* b, float* c, int M, int K, int N) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += (double)a[(size_t)m * K + k] * (double)b[(size_t)n * K + k];
            c[(size_t)m * N + n] = (float)s;
        }
}

/* softmax over the last axis of a [rows][n] matrix, max-subtracted */
void tdc_softmax_lastdim(const float* x, float* y, int rows, int n) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* p = x + (size_t)r * n;
        float m = p[0];
        for (int i = 1; i < n; ++i) if (p[i] > m) m = p[i];
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += exp((double)p[i] - (double)m);
        for (int i = 0; i < n; ++i) y[(size_t)r * n + i] = (float)(exp((double)p[i] - (double)m) / s);
    }
}

/* per channel plane: y = (x - mean) / sqrt(var + eps) * g[p] + b[p], var biased, g/b of the plane's shape shared by channels */
void tdc_layer_norm_plane(const float* x, int C, int HW, const float* g, const float* b, double eps, float* y) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float* p = x + (size_t)c * HW;
        double s = 0.0;
        for (int i = 0; i < HW; ++i) s += (double)p[i];
        const double mean = s / HW;
        double v = 0.0;
        for (int i = 0; i < HW; ++i) { const double d = (double)p[i] - mean; v += d * d; }
        const double rstd = 1.0 / sqrt(v / HW + eps);
        for (int i = 0; i < HW; ++i) y[(size_t)c * HW + i] = (float)(((double)p[i] - mean) * rstd * (double)g[i] + (double)b[i]);
    }
}
