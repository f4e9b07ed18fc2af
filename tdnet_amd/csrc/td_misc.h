// td_misc.h -- the HBM-bound tail of the TDNet hot path: layout change, max-pool, pyramid pooling, plane LayerNorm,
// classifier, bilinear upsample, argmax, key/value sub-sampling.  All NHWC fp32, float4 per lane, grid-stride.
#pragma once
#include "td_device.h"
#include "td_conv.h"   // td_ld4 / td_st4

static inline int td_grid_for(long work_items, int block = 256, int max_blocks = 256 * 8) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > max_blocks ? max_blocks : g);
}

// ---- image NCHW [3][H][W] -> NHWC4 [H][W][4] (4th channel 0) so the stem gathers one float4 per tap --------------
TD_KERNEL void k_nchw3_to_nhwc4(const float* __restrict__ img, float* __restrict__ out, int HW) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        f32x4 v = {img[p], img[HW + p], img[2 * HW + p], 0.f};
        td_st4(out + (size_t)p * 4, v);
    }
}

// the same, 4 consecutive pixels per thread: three 16-byte loads (one per colour plane) and four 16-byte stores (HW % 4 == 0)
TD_KERNEL void k_nchw3_to_nhwc4_x4(const float* __restrict__ img, float* __restrict__ out, int HW) {
    const int n4 = HW >> 2;
    for (int p4 = blockIdx.x * blockDim.x + threadIdx.x; p4 < n4; p4 += gridDim.x * blockDim.x) {
        const f32x4 r = td_ld4(img + (size_t)p4 * 4), g = td_ld4(img + HW + (size_t)p4 * 4), b = td_ld4(img + 2 * (size_t)HW + (size_t)p4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 v = {r[e], g[e], b[e], 0.f};
            td_st4(out + ((size_t)p4 * 4 + e) * 4, v);
        }
    }
}

// ---- image NCHW [3][H][W] -> the packed-row image of the 7x7 stem: out[(y + 3) Wp + x + 4][3] (td_conv_ad.h STEM = 2; Wp % 4 == 0).  The
// border (zeros) is written once, when the workspace is allocated.  A thread moves 4 consecutive pixels of a row: three 16-byte loads
// (one per colour plane, where the plane rows are 16-byte aligned) and three 16-byte stores of 12 contiguous, aligned floats.
TD_KERNEL void k_nchw3_to_rgbpad(const float* __restrict__ img, float* __restrict__ out, int H, int W, int Wp) {
    const int W4 = (W + 3) >> 2;
    const long total = (long)H * W4;
    const long HW = (long)H * W;
    const bool vec = (W & 3) == 0 && (((size_t)img) & 15) == 0;       // uniform
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W4), x0 = (int)(i % W4) * 4;
        const long p = (long)y * W + x0;
        float* o = out + ((size_t)(y + 3) * Wp + x0 + 4) * 3;         // 16-byte aligned: Wp % 4 == 0, x0 % 4 == 0
        if (vec) {
            const f32x4 r = td_ld4(img + p), g = td_ld4(img + HW + p), b = td_ld4(img + 2 * HW + p);
            const f32x4 v0 = {r[0], g[0], b[0], r[1]}, v1 = {g[1], b[1], r[2], g[2]}, v2 = {b[2], r[3], g[3], b[3]};
            td_st4(o, v0); td_st4(o + 4, v1); td_st4(o + 8, v2);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < W) { o[3 * e] = img[p + e]; o[3 * e + 1] = img[HW + p + e]; o[3 * e + 2] = img[2 * HW + p + e]; }
        }
    }
}

// ---- MaxPool2d(3, stride 2, pad 1), padding = -inf, floor mode (resnet.py:137) -----------------------------------
TD_KERNEL void k_maxpool3s2(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int Ho, int Wo) {
    const int CV = C >> 2;
    const long total = (long)Ho * Wo * CV;
    // nine UNCONDITIONAL range-checked loads, all in flight, padding taps replaced by -3e38 with a select: `if (inside) load; max`
    // is nine dependent memory round trips per output (59 us for 168 MB at 1024x2048)
    const TdBuf in_buf = td_make_buf(in, (unsigned)H * (unsigned)W * (unsigned)C * 4u);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long pix = i / CV;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        f32x4 v[3][3];
        bool ok[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                ok[ky][kx] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                v[ky][kx] = td_buf_ld4(in_buf, ok[ky][kx] ? (((unsigned)iy * (unsigned)W + (unsigned)ix) * (unsigned)C + (unsigned)cv * 4u) * 4u : TD_BUF_OOB, 0u);
            }
        f32x4 m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = ok[ky][kx] ? v[ky][kx][e] : -3.0e38f;
                    m[e] = t > m[e] ? t : m[e];
                }
        td_st4(out + (size_t)pix * C + cv * 4, m);
    }
}

// the same for the fp16-activation mode (tdnet_opts.precision = 1): the pooled map is stored as fp16 (the first map of the fp16
// backbone); the input is the stem's fp32 output (IN16 = false) or, behind the deep stem, already fp16 (IN16 = true)
typedef _Float16 td_f16x4 __attribute__((ext_vector_type(4)));
template <bool IN16>
TD_KERNEL void k_maxpool3s2_h(const void* __restrict__ inv, _Float16* __restrict__ out, int H, int W, int C, int Ho, int Wo) {
    const int CV = C >> 2;
    const long total = (long)Ho * Wo * CV;
    const TdBuf in_buf = td_make_buf(reinterpret_cast<const float*>(inv), (unsigned)H * (unsigned)W * (unsigned)C * (IN16 ? 2u : 4u));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long pix = i / CV;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        f32x4 v[3][3];                                                // as k_maxpool3s2: all nine taps in flight, padding by select
        bool ok[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                ok[ky][kx] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const unsigned el = ((unsigned)iy * (unsigned)W + (unsigned)ix) * (unsigned)C + (unsigned)cv * 4u;
                if (IN16) {
                    const td_f16x4 h = __builtin_bit_cast(td_f16x4, td_buf_ld2(in_buf, ok[ky][kx] ? el * 2u : TD_BUF_OOB, 0u));
                    v[ky][kx] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                } else v[ky][kx] = td_buf_ld4(in_buf, ok[ky][kx] ? el * 4u : TD_BUF_OOB, 0u);
            }
        f32x4 m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = ok[ky][kx] ? v[ky][kx][e] : -3.0e38f;
                    m[e] = t > m[e] ? t : m[e];
                }
        td_f16x4 oh = {(_Float16)m[0], (_Float16)m[1], (_Float16)m[2], (_Float16)m[3]};
        *reinterpret_cast<td_f16x4*>(out + (size_t)pix * C + cv * 4) = oh;
    }
}

// the same, two horizontally adjacent outputs per thread: they share the middle input column, 15 loads for 2 outputs instead of 18
TD_KERNEL void k_maxpool3s2_x2(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int Ho, int Wo) {
    const int CV = C >> 2, Wp = (Wo + 1) >> 1;
    const long total = (long)Ho * Wp * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long pp = i / CV;
        const int oxp = (int)(pp % Wp), oy = (int)(pp / Wp);
        const int ox0 = 2 * oxp;
        const f32x4 NEG4 = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        f32x4 m0 = NEG4, m1 = NEG4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            f32x4 c[5];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int ix = 2 * ox0 - 1 + kx;
                c[kx] = (unsigned)ix < (unsigned)W ? td_ld4(in + ((size_t)iy * W + ix) * C + cv * 4) : NEG4;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = c[0][e] > c[1][e] ? c[0][e] : c[1][e];
                a = c[2][e] > a ? c[2][e] : a;
                float b = c[3][e] > c[4][e] ? c[3][e] : c[4][e];
                b = c[2][e] > b ? c[2][e] : b;
                m0[e] = a > m0[e] ? a : m0[e];
                m1[e] = b > m1[e] ? b : m1[e];
            }
        }
        td_st4(out + ((size_t)oy * Wo + ox0) * C + cv * 4, m0);
        if (ox0 + 1 < Wo) td_st4(out + ((size_t)oy * Wo + ox0 + 1) * C + cv * 4, m1);
    }
}

// ---- Pyramid pooling (td4_psp18.py:243-284) ----------------------------------------------------------------------
// AdaptiveAvgPool2d(o): bin i covers [floor(i n / o), ceil((i+1) n / o)).  12 x-bins (levels 1,2,3,6) per row first,
// then rows are combined per y-bin: 50 bins x C sums, each input element read once per level.
TD_HOSTDEV int td_bin_lo(int i, int n, int o) { return (i * n) / o; }
TD_HOSTDEV int td_bin_hi(int i, int n, int o) { return ((i + 1) * n + o - 1) / o; }

// Row sums of the pyramid.  The 12 x-bins of a row (levels 1, 2, 3, 6; adaptive pooling: [floor(i w / o), ceil((i + 1) w / o)), so
// neighbours overlap by a column) cover the row four times; summing each bin on its own read the map 4x (49 us for 67 MB at
// 1024x2048, and the level-1 bin -- a whole row -- was one 256-long chain).  The bin edges of all levels cut the row into at most 23
// ATOMS (10 at w = 256); every atom is summed once and every bin is a run of consecutive atoms (k_ppm_bins).
struct PpmAtoms {
    int n;                // atoms per row
    int edge[25];         // atom a = columns [edge[a], edge[a + 1])
    int lo[12], hi[12];   // x-bin b (level-major: 1 | 2 | 3 | 6) = atoms lo[b] .. hi[b] - 1
};
static inline PpmAtoms ppm_atoms(int w) {
    static const int lv[4] = {1, 2, 3, 6};
    PpmAtoms a;
    int pts[24], np = 0;
    for (int l = 0; l < 4; ++l)
        for (int i = 0; i < lv[l]; ++i) { pts[np++] = td_bin_lo(i, w, lv[l]); pts[np++] = td_bin_hi(i, w, lv[l]); }
    for (int i = 1; i < np; ++i)                                     // insertion sort, then unique
        for (int j = i; j > 0 && pts[j - 1] > pts[j]; --j) { const int t = pts[j]; pts[j] = pts[j - 1]; pts[j - 1] = t; }
    int ne = 0;
    for (int i = 0; i < np; ++i) if (ne == 0 || a.edge[ne - 1] != pts[i]) a.edge[ne++] = pts[i];
    a.n = ne - 1;
    for (int i = ne; i < 25; ++i) a.edge[i] = a.edge[ne - 1];
    int b = 0;
    for (int l = 0; l < 4; ++l)
        for (int i = 0; i < lv[l]; ++i, ++b) {
            const int lo = td_bin_lo(i, w, lv[l]), hi = td_bin_hi(i, w, lv[l]);
            a.lo[b] = a.hi[b] = 0;
            for (int e = 0; e < ne; ++e) { if (a.edge[e] == lo) a.lo[b] = e; if (a.edge[e] == hi) a.hi[b] = e; }
        }
    return a;
}
// One launch per ROW GROUP (round 5; rounds 2-4: k_ppm_rowsum + k_ppm_rowbins, two launches through a [h][atoms][C] buffer in HBM).
// grid = h rows x C / 512 channel groups, block = 1024 = 128 channel vectors x 8 atom slots: thread (cv, as) sums the atoms as, as + 8,
// ... of its row for its four channels (an atom = a run of columns, summed left to right) into LDS, then the 12 x-bins of the row are
// runs of consecutive atoms, added in atom order.  Same sums in the same order as the two kernels it replaces: bit-identical rowbins.
TD_KERNEL void TD_LAUNCH_BOUNDS(1024, 1) k_ppm_rowbins(const float* __restrict__ c4, float* __restrict__ rowbins, int w, int C, PpmAtoms at) {
    TD_DYN_LDS(smem);
    float* atoms = reinterpret_cast<float*>(smem);                 // [at.n][512]
    const int groups = (C + 511) / 512, y = blockIdx.x / groups, cg = blockIdx.x % groups;
    const int cv = threadIdx.x & 127, as = threadIdx.x >> 7, c = cg * 512 + cv * 4;
    const bool cok = c < C;
    const float* row = c4 + (size_t)y * w * C + c;
    for (int a = as; a < at.n; a += 8) {
        const int lo = at.edge[a], hi = at.edge[a + 1];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (cok) {
#pragma unroll 8
            for (int x = lo; x < hi; ++x) s = s + td_ld4(row + (size_t)x * C);
        }
        td_st4(atoms + a * 512 + cv * 4, s);
    }
    __syncthreads();
    for (int b = as; b < 12; b += 8) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int a = at.lo[b]; a < at.hi[b]; ++a) s = s + td_ld4(atoms + a * 512 + cv * 4);
        if (cok) td_st4(rowbins + ((size_t)y * 12 + b) * C + c, s);
    }
}
// Pooling + pyramid 1x1 conv in one launch (round 5; rounds 2-4: k_ppm_bins + k_ppm_conv through a [50][C] buffer in HBM).
// grid = 50 bins x FS / 64 channel groups, block = 256.  Phase 1: the bin's mean over its rows of the row bins (bin order: level-major,
// then by, then bx; rows added top to bottom) into LDS.  Phase 2: the pyramid conv (BN folded) + ReLU on that vector, only the FS channels
// this path keeps: feat[bin][f] = relu(b[lvl][f] + sum_c W[lvl][c][f] pooled[c]) -- thread (f, slice) sums a quarter of the input channels
// with 8 loads in flight, LDS combines the four slices in a fixed order.  Same arithmetic as the two kernels it replaces: bit-identical feat.
TD_KERNEL void k_ppm_pool_conv(const float* __restrict__ rowbins, const float* __restrict__ wgt, const float* __restrict__ bias,
                               float* __restrict__ feat, int h, int w, int C, int FS) {
    TD_DYN_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);               // [4][64]
    float* pv = red + 256;                                      // [C]: the pooled vector of this bin
    const int groups = FS >> 6, bin = blockIdx.x / groups, fl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int f = (blockIdx.x % groups) * 64 + fl;
    const int lvl = bin >= 14 ? 3 : bin >= 5 ? 2 : bin >= 1 ? 1 : 0;
    {
        const int o = lvl == 0 ? 1 : lvl == 1 ? 2 : lvl == 2 ? 3 : 6, xoff = lvl == 0 ? 0 : lvl == 1 ? 1 : lvl == 2 ? 3 : 6;
        const int lb = bin - (lvl == 0 ? 0 : lvl == 1 ? 1 : lvl == 2 ? 5 : 14);
        const int by = lb / o, bx = lb % o;
        const int ylo = td_bin_lo(by, h, o), yhi = td_bin_hi(by, h, o);
        const int cnt = (yhi - ylo) * (td_bin_hi(bx, w, o) - td_bin_lo(bx, w, o));
        for (int cv = threadIdx.x; cv < (C >> 2); cv += blockDim.x) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 32                                               // the level-0 bin adds every row of the map: 32 loads in flight per trip, same order
            for (int y = ylo; y < yhi; ++y) s = s + td_ld4(rowbins + ((size_t)y * 12 + xoff + bx) * C + cv * 4);
            td_st4(pv + cv * 4, s * (1.0f / (float)cnt));
        }
    }
    __syncthreads();
    const int cq = C >> 2;
    const float* wr = wgt + ((size_t)lvl * C + (size_t)sl * cq) * FS + f;   // weights stored [lvl][c][f]: lanes read consecutive f
    const float* pp = pv + sl * cq;
    float s = 0.f;
    // one dependent fma chain, its loads 32 at a time (8 or 32 at a time: 13.3 us per launch at 720x960 either way).  Requesting all 128
    // weights of the thread BEFORE the pooling phase was measured too: 28.6 us -- with 128 registers held the compiler serialises the pooling
    // loop's loads (and under the default 1024-thread bound it spilled: 45 us), profiles/r05m_*.
#pragma unroll 32
    for (int c = 0; c < cq; ++c) s = fmaf(wr[(size_t)c * FS], pp[c], s);
    red[sl * 64 + fl] = s;
    __syncthreads();
    if (sl == 0) {
        s = ((red[fl] + red[64 + fl]) + red[128 + fl]) + red[192 + fl] + bias[lvl * FS + f];
        feat[(size_t)bin * FS + f] = s > 0.f ? s : 0.f;
    }
}
// z[p] = [ c4[p][pid*XS : +XS] | bilinear(feat_l)(p)[0:FS] for l = 0..3 ]   (align_corners=True, td4_psp18.py:273-284)
TD_KERNEL void k_ppm_assemble(const float* __restrict__ c4, const float* __restrict__ feat, float* __restrict__ z,
                              int h, int w, int C, int xs_off, int XS, int FS) {
    const int ZC = XS + 4 * FS, ZV = ZC >> 2;
    const long total = (long)h * w * ZV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int zv = (int)(i % ZV);
        const long pix = i / ZV;
        const int c = zv * 4;
        f32x4 v;
        if (c < XS) {
            v = td_ld4(c4 + (size_t)pix * C + xs_off + c);
        } else {
            const int lvl = (c - XS) / FS, f = (c - XS) % FS;
            const int o = lvl == 0 ? 1 : lvl == 1 ? 2 : lvl == 2 ? 3 : 6;
            const int boff = lvl == 0 ? 0 : lvl == 1 ? 1 : lvl == 2 ? 5 : 14;
            const int x = (int)(pix % w), y = (int)(pix / w);
            const float sy = (h > 1) ? (float)(o - 1) / (float)(h - 1) : 0.f;
            const float sx = (w > 1) ? (float)(o - 1) / (float)(w - 1) : 0.f;
            const float fy = sy * (float)y, fx = sx * (float)x;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < o - 1 ? 1 : 0), x1 = x0 + (x0 < o - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float* fb = feat + (size_t)boff * FS + f;
            const f32x4 v00 = td_ld4(fb + (size_t)(y0 * o + x0) * FS), v01 = td_ld4(fb + (size_t)(y0 * o + x1) * FS);
            const f32x4 v10 = td_ld4(fb + (size_t)(y1 * o + x0) * FS), v11 = td_ld4(fb + (size_t)(y1 * o + x1) * FS);
            v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        td_st4(z + (size_t)pix * ZC + c, v);
    }
}

// ---- LayerNorm over the (h,w) plane of each channel (td4_psp18.py:306-312), NHWC ----------------------------------
// Two-pass (centred) variance without reading the plane twice from HBM: a strip of pixels is swept twice by ONE workgroup
// (the second sweep hits L2), which yields the strip's mean m_s and M2_s = sum (x - m_s)^2; the strips are then combined
// exactly:  mean = sum n_s m_s / N,   var = sum (M2_s + n_s (m_s - mean)^2) / N.   Fixed summation order everywhere.
// pass A: part[s][C] = m_s, part[nstr + s][C] = M2_s  (empty strips write zeros)
TD_KERNEL void k_ln_stats(const float* __restrict__ x, float* __restrict__ part, int HW, int C) {
    TD_DYN_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);               // [rows][C], then [C] strip means behind it
    const int CV = C >> 2, rows = blockDim.x / CV;
    float* smean = red + (size_t)rows * C;
    const int cv = threadIdx.x % CV, r = threadIdx.x / CV;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const int cnt = p1 > p0 ? p1 - p0 : 0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + r; p < p1; p += rows) s = s + td_ld4(x + (size_t)p * C + cv * 4);
    td_st4(red + (size_t)r * C + cv * 4, s);
    __syncthreads();
    if (r == 0) {
        for (int k = 1; k < rows; ++k) s = s + td_ld4(red + (size_t)k * C + cv * 4);
        const f32x4 m = cnt ? s * (1.0f / (float)cnt) : s;
        td_st4(smean + cv * 4, m);
        td_st4(part + (size_t)blockIdx.x * C + cv * 4, m);
    }
    __syncthreads();
    const f32x4 m = td_ld4(smean + cv * 4);
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + r; p < p1; p += rows) {
        const f32x4 v = td_ld4(x + (size_t)p * C + cv * 4) - m;
        q = q + v * v;
    }
    td_st4(red + (size_t)r * C + cv * 4, q);
    __syncthreads();
    if (r == 0) {
        for (int k = 1; k < rows; ++k) q = q + td_ld4(red + (size_t)k * C + cv * 4);
        td_st4(part + ((size_t)gridDim.x + blockIdx.x) * C + cv * 4, q);
    }
}
// pass B: mean[c] and rstd[c] = 1/sqrt(var + eps) from the strip statistics; strip k covers rows [k*per, min((k+1)*per, HW)).
// grid = C/4, block = 256 = 4 channels x 64 strip slices (a slice walks strips sl, sl + 64, ...); the 64 slice partials of a
// channel are combined 8 x 8 in a fixed order (deterministic).  (The first version ran 16 channels x 16 slices on C/16 = 32
// workgroups: 64 dependent iterations per phase, 22 us for 4 MB of statistics.)
TD_KERNEL void k_ln_finalize(const float* __restrict__ part, int nstrips, int per, int HW, int C, float eps, float* __restrict__ mean,
                             float* __restrict__ rstd) {
    TD_DYN_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);               // [64 slices][4 channels], [8][4] second level, [4] means
    float* red2 = red + 256;
    float* mu = red2 + 32;
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    auto count = [&](int k) { const int p0 = k * per, p1 = (p0 + per < HW) ? p0 + per : HW; return p1 > p0 ? (float)(p1 - p0) : 0.f; };
    auto combine = [&](float v) -> float {                      // sum over the 64 slices of channel cl, identical on the block's first 4 threads
        red[sl * 4 + cl] = v;
        __syncthreads();
        if (sl < 8) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[(sl * 8 + k) * 4 + cl];
            red2[sl * 4 + cl] = t;
        }
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red2[k * 4 + cl];
        __syncthreads();
        return t;
    };
    float s = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int k = sl; k < nstrips; k += 64) s += count(k) * part[(size_t)k * C + c];
    }
    const float m = combine(s) / (float)HW;
    s = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int k = sl; k < nstrips; k += 64) {
            const float d = part[(size_t)k * C + c] - m;
            s += part[((size_t)nstrips + k) * C + c] + count(k) * d * d;
        }
    }
    const float t = combine(s);
    (void)mu;
    if (sl == 0 && c < C) {
        mean[c] = m;
        rstd[c] = 1.0f / sqrtf(t / (float)HW + eps);
    }
}
// pass C: y = (x - mean[c]) * rstd[c] * g[p] + b[p]
TD_KERNEL void k_ln_apply(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                          const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ y, int HW, int C) {
    const int CV = C >> 2;
    const long total = (long)HW * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long p = i / CV;
        const f32x4 v = td_ld4(x + (size_t)p * C + cv * 4);
        const f32x4 m = td_ld4(mean + cv * 4), rs = td_ld4(rstd + cv * 4);
        td_st4(y + (size_t)p * C + cv * 4, (v - m) * rs * g[p] + b[p]);
    }
}

// The same with an fp16 map out (tdnet_opts.precision = 1: the head's 3x3 conv reads fp16 through the LDS-DMA kernel).  The fp32 value
// is the one k_ln_apply computes and the rounding is the one the conv kernel applies to an fp32 input when it stages it.
TD_KERNEL void k_ln_apply_h(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                            const float* __restrict__ g, const float* __restrict__ b, _Float16* __restrict__ y, int HW, int C) {
    const int CV = C >> 2;
    const long total = (long)HW * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long p = i / CV;
        const f32x4 v = td_ld4(x + (size_t)p * C + cv * 4);
        const f32x4 m = td_ld4(mean + cv * 4), rs = td_ld4(rstd + cv * 4);
        const f32x4 r = (v - m) * rs * g[p] + b[p];
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (_Float16)r[k];
        *reinterpret_cast<f16x4*>(y + (size_t)p * C + cv * 4) = o;
    }
}

// ---- classifier: 1x1 conv C -> NC (+bias) (td4_psp18.py:299), NHWC in, PLANAR [NC][HW] out ------------------------
// One workgroup = 64 pixels x 4 channel quarters (one wave each): a lane streams its quarter of its pixel's channels (C bytes,
// whole cache lines) against the LDS-resident weights, the four partial sums meet in LDS and are added in a fixed order.
// (One thread per pixel, the first version, left three quarters of the SIMDs without a wave at 128x256: 40 us for 17 MB.)
template <int NC_MAX>
TD_KERNEL void k_classifier(const float* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                            float* __restrict__ out, int HW, int C, int NC) {
    TD_DYN_LDS(smem);
    float* ws = reinterpret_cast<float*>(smem);                // [NC][C]
    float* red = ws + NC * C;                                  // [4][NC][64]
    for (int i = threadIdx.x; i < NC * C; i += blockDim.x) ws[i] = wgt[i];
    __syncthreads();
    const int lp = threadIdx.x & 63, q = threadIdx.x >> 6, CQ = C >> 2;
    const int p = blockIdx.x * 64 + lp;
    float acc[NC_MAX];
#pragma unroll
    for (int k = 0; k < NC_MAX; ++k) acc[k] = 0.f;
    if (p < HW) {
        const float* xp = x + (size_t)p * C + q * CQ;
#pragma unroll 4
        for (int c = 0; c < CQ; c += 4) {
            const f32x4 v = td_ld4(xp + c);
#pragma unroll
            for (int k = 0; k < NC_MAX; ++k) {
                if (k < NC) {
                    const float* wr = ws + k * C + q * CQ + c;
                    acc[k] = fmaf(v[0], wr[0], acc[k]); acc[k] = fmaf(v[1], wr[1], acc[k]);
                    acc[k] = fmaf(v[2], wr[2], acc[k]); acc[k] = fmaf(v[3], wr[3], acc[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC_MAX; ++k)
        if (k < NC) red[(q * NC + k) * 64 + lp] = acc[k];
    __syncthreads();
    for (int v = threadIdx.x; v < NC * 64; v += blockDim.x) {
        const int k = v >> 6, pl = v & 63, pp = blockIdx.x * 64 + pl;
        if (pp >= HW) continue;
        const float sum = ((red[(0 * NC + k) * 64 + pl] + red[(1 * NC + k) * 64 + pl]) + red[(2 * NC + k) * 64 + pl]) + red[(3 * NC + k) * 64 + pl];
        out[(size_t)k * HW + pp] = sum + bias[k];
    }
}

// ---- bilinear, align_corners=True (td4_psp18.py:227): planar [C][h][w] -> [C][H][W] ------------------------------
struct UpCoef { int i0, i1; float l; };
TD_DEV UpCoef td_up_coef(int d, float scale, int n_in) {
    const float f = scale * (float)d;
    UpCoef c;
    c.i0 = (int)f;
    c.i1 = c.i0 + (c.i0 < n_in - 1 ? 1 : 0);
    c.l = f - (float)c.i0;
    return c;
}
TD_KERNEL void k_upsample(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w, int H, int W) {
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = (long)C * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W);
        const long t = i / W;
        const int Y = (int)(t % H), c = (int)(t / H);
        const UpCoef cy = td_up_coef(Y, sy, h), cx = td_up_coef(X, sx, w);
        const float* pl = in + (size_t)c * h * w;
        const float v00 = pl[cy.i0 * w + cx.i0], v01 = pl[cy.i0 * w + cx.i1];
        const float v10 = pl[cy.i1 * w + cx.i0], v11 = pl[cy.i1 * w + cx.i1];
        out[i] = (1.f - cy.l) * ((1.f - cx.l) * v00 + cx.l * v01) + cy.l * ((1.f - cx.l) * v10 + cx.l * v11);
    }
}
// Any W (769x1537, the reference's native size, is not a multiple of 4, so the rows of the [C][H][W] output start at every alignment).
// grid = (ceil((W / 4 + 2) / 256), H, C): row and channel from the block index, vertical coefficients wave-uniform.  Lane q >= 1 of a row
// writes the 16-byte ALIGNED quad X0 + 4 (q - 1) .. + 3 with one store, X0 = the row's first aligned column; lane 0 writes the X0 head
// elements, the lane of the last (partial) quad its tail, as scalars.  Same expression per element as k_upsample: bit-identical.
// (Round 5: the grid-stride k_upsample with a 64-bit div / mod per element took 70 us for the 90 MB of a 769x1537 frame, one 4-byte store
// per lane 48 us.)
TD_KERNEL void k_upsample_row(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w, int H, int W) {
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int q = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y, c = blockIdx.z;
    float* orow = out + ((size_t)c * H + Y) * W;
    const int X0 = (int)((4u - (unsigned)(((size_t)orow >> 2) & 3u)) & 3u);      // columns before the first 16-byte boundary of this row
    const int xa = q == 0 ? 0 : X0 + 4 * (q - 1), xb = q == 0 ? (X0 < W ? X0 : W) : (xa + 4 < W ? xa + 4 : W);
    if (xa >= xb) return;
    const UpCoef cy = td_up_coef(Y, sy, h);
    const float* r0 = in + ((size_t)c * h + cy.i0) * w;
    const float* r1 = in + ((size_t)c * h + cy.i1) * w;
    auto value = [&](int X) {
        const UpCoef cx = td_up_coef(X, sx, w);
        return (1.f - cy.l) * ((1.f - cx.l) * r0[cx.i0] + cx.l * r0[cx.i1]) + cy.l * ((1.f - cx.l) * r1[cx.i0] + cx.l * r1[cx.i1]);
    };
    if (q > 0 && xb - xa == 4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = value(xa + e);
        td_st4(orow + xa, o);
    } else {
        for (int X = xa; X < xb; ++X) orow[X] = value(X);
    }
}
// same arithmetic, 4 consecutive output columns per lane and one 16-byte store (W % 4 == 0): the 159 MB logits write
// of a 1024x2048 frame is the largest single HBM stream of the path
// grid = (ceil(W/4 / 256), H, C): the row and channel come from the block index (no 64-bit div/mod per thread), the row's
// vertical coefficients are wave-uniform
TD_KERNEL void k_upsample_x4(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w, int H, int W) {
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int X4 = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y, c = blockIdx.z;
    if (X4 >= (W >> 2)) return;
    const UpCoef cy = td_up_coef(Y, sy, h);
    const float* r0 = in + ((size_t)c * h + cy.i0) * w;
    const float* r1 = in + ((size_t)c * h + cy.i1) * w;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const UpCoef cx = td_up_coef(X4 * 4 + e, sx, w);
        o[e] = (1.f - cy.l) * ((1.f - cx.l) * r0[cx.i0] + cx.l * r0[cx.i1]) + cy.l * ((1.f - cx.l) * r1[cx.i0] + cx.l * r1[cx.i1]);
    }
    td_st4(out + ((size_t)c * H + Y) * W + X4 * 4, o);
}
// argmax over classes, first maximum wins (== output.max(1)[1], test.py:61); labels int32 [H][W]
TD_KERNEL void k_argmax(const float* __restrict__ logits, int32_t* __restrict__ labels, int C, long HW) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        float best = logits[p];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = logits[(size_t)c * HW + p];
            if (v > best) { best = v; bi = c; }
        }
        labels[p] = bi;
    }
}
// fused upsample + argmax: the same arithmetic as k_upsample followed by k_argmax, without the [C][H][W] round trip
TD_KERNEL void k_upsample_argmax(const float* __restrict__ in, int32_t* __restrict__ labels, int C, int h, int w, int H, int W) {
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)(i / W);
        const UpCoef cy = td_up_coef(Y, sy, h), cx = td_up_coef(X, sx, w);
        float best = 0.f;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float* pl = in + (size_t)c * h * w;
            const float v00 = pl[cy.i0 * w + cx.i0], v01 = pl[cy.i0 * w + cx.i1];
            const float v10 = pl[cy.i1 * w + cx.i0], v11 = pl[cy.i1 * w + cx.i1];
            const float v = (1.f - cy.l) * ((1.f - cx.l) * v00 + cx.l * v01) + cy.l * ((1.f - cx.l) * v10 + cx.l * v11);
            if (c == 0 || v > best) { best = v; bi = c; }
        }
        labels[i] = bi;
    }
}

// ---- stride-4 sub-sampling of an NHWC map (MaxPool2d(kernel 1, stride 4), transformer.py:26,36) -------------------
TD_KERNEL void k_subsample(const float* __restrict__ in, float* __restrict__ out, int w, int C, int ho, int wo, int stride) {
    const int CV = C >> 2;
    const long total = (long)ho * wo * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long pix = i / CV;
        const int ox = (int)(pix % wo), oy = (int)(pix / wo);
        td_st4(out + (size_t)pix * C + cv * 4, td_ld4(in + ((size_t)(oy * stride) * w + ox * stride) * C + cv * 4));
    }
}
// Both cache entries of a frame in ONE launch (round 5): q_ = q_cur[::4, ::4] (C1 channels) and v_ = v_cur[::4, ::4] (C2 channels) --
// Encoding(pre=True)'s q_ and v_ are the stride-4 sub-sample of the full-resolution projections (transformer.py:34-50).
TD_KERNEL void k_subsample2(const float* __restrict__ in1, float* __restrict__ out1, int C1, const float* __restrict__ in2,
                            float* __restrict__ out2, int C2, int w, int ho, int wo, int stride) {
    const int CV1 = C1 >> 2, CV = (C1 + C2) >> 2;
    const long total = (long)ho * wo * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long pix = i / CV;
        const int ox = (int)(pix % wo), oy = (int)(pix / wo);
        const size_t src = (size_t)(oy * stride) * w + ox * stride;
        if (cv < CV1) td_st4(out1 + (size_t)pix * C1 + cv * 4, td_ld4(in1 + src * C1 + cv * 4));
        else td_st4(out2 + (size_t)pix * C2 + (cv - CV1) * 4, td_ld4(in2 + src * C2 + (cv - CV1) * 4));
    }
}
// fp32 <-> fp16 copies (test entry tdnet_op_conv2d_f16io only: the product path never converts whole maps)
TD_KERNEL void k_f2h(const float* __restrict__ in, _Float16* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (_Float16)in[i];
}
TD_KERNEL void k_h2f(const _Float16* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
// NHWC [HW][C] -> planar [C][HW] into HOST-layout staging (used only by tdnet_get_stage)
TD_KERNEL void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, long HW, int C) {
    const long total = HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i % HW;
        const int c = (int)(i / HW);
        out[i] = in[(size_t)p * C + c];
    }
}
