// td_wino.h -- Winograd F(2x2, 3x3) for the stride-1 dilated 3x3 convolutions of layers 3-4 (fp32).
//
// Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A  turns every 2x2 output tile into 16 products per (ci, co) instead of 36:
// the contraction shrinks 2.25x and becomes 16 independent [tiles x Cin] x [Cin x Cout] GEMMs, which run on the same
// fp32-MFMA kernel as the 1x1 convs (k_conv_igemm with nbatch = 16).  The input/output transforms are HBM-bound passes.
//
// A conv with dilation d (resnet.py:32-37: 2, 4, 8, 16 here) is d*d independent dilation-1 convs on the sub-grids
// {(py + d a, px + d b)}: a tile is (phase py, px; tile ty, tx) and its 4x4 input patch is read with stride d.
//   tile index t = ((py*d + px) * TY + ty) * TX + tx,  TY = ceil(ceil(H/d)/2), TX = ceil(ceil(W/d)/2)
// so all phases have the same tile count (out-of-range taps read zeros, out-of-range outputs are not written).
//
// Numerics: fp32 throughout; the transforms add a few roundings per element (error ~4x a direct fp32 conv, still 1e-7
// relative).  Opt-in per DESIGN.md: it changes the summation structure, not the precision.
#pragma once
#include "td_conv.h"

#include <vector>

struct WinoArgs {
    const float* in;      // [H][W][C]
    float* V;             // [16][T][C]
    const float* Mb;      // [16][T][Cout]
    const float* bias;    // [Cout]
    const float* resid;   // [H][W][Cout] or nullptr
    float* out;           // [H][W][Cout]
    int H, W, C, Cout, dil, TY, TX, T, act;
};

// B^T d B for one 4x4 patch held as d[r][c] (each a float4 of channels)
TD_DEV void td_wino_bt_d_b(const f32x4 (&d)[4][4], f32x4 (&v)[4][4]) {
    f32x4 t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {                                   // rows: B^T d
        t[0][c] = d[0][c] - d[2][c];
        t[1][c] = d[1][c] + d[2][c];
        t[2][c] = d[2][c] - d[1][c];
        t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                   // columns: (.) B
        v[r][0] = t[r][0] - t[r][2];
        v[r][1] = t[r][1] + t[r][2];
        v[r][2] = t[r][2] - t[r][1];
        v[r][3] = t[r][1] - t[r][3];
    }
}

// thread = (tile, 4 channels); lanes run over channels (coalesced float4)
TD_KERNEL void k_wino_in(WinoArgs p) {
    const int CV = p.C >> 2;
    const long total = (long)p.T * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int t = (int)(i / CV);
        const int tx = t % p.TX; t /= p.TX;
        const int ty = t % p.TY; t /= p.TY;
        const int px = t % p.dil, py = t / p.dil;
        f32x4 d[4][4], v[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = py + p.dil * (2 * ty - 1 + r);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int x = px + p.dil * (2 * tx - 1 + c);
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) z = td_ld4(p.in + ((size_t)y * p.W + x) * p.C + cv * 4);
                d[r][c] = z;
            }
        }
        td_wino_bt_d_b(d, v);
        const size_t tile = (size_t)(i / CV);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) td_st4(p.V + ((size_t)(r * 4 + c) * p.T + tile) * p.C + cv * 4, v[r][c]);
    }
}

// thread = (tile, 4 output channels): Y = A^T m A, + bias (+ residual), activation, scatter to the 2x2 output pixels
TD_KERNEL void k_wino_out(WinoArgs p) {
    const int CV = p.Cout >> 2;
    const long total = (long)p.T * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const size_t tile = (size_t)(i / CV);
        int t = (int)tile;
        const int tx = t % p.TX; t /= p.TX;
        const int ty = t % p.TY; t /= p.TY;
        const int px = t % p.dil, py = t / p.dil;
        f32x4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = td_ld4(p.Mb + ((size_t)(r * 4 + c) * p.T + tile) * p.Cout + cv * 4);
        f32x4 s[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                // A^T m
            s[0][c] = m[0][c] + m[1][c] + m[2][c];
            s[1][c] = m[1][c] - m[2][c] - m[3][c];
        }
        const f32x4 b = td_ld4(p.bias + cv * 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = py + p.dil * (2 * ty + r);
            if (y >= p.H) continue;
            f32x4 o2[2];
            o2[0] = s[r][0] + s[r][1] + s[r][2];                      // (.) A
            o2[1] = s[r][1] - s[r][2] - s[r][3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int x = px + p.dil * (2 * tx + c);
                if (x >= p.W) continue;
                const size_t off = ((size_t)y * p.W + x) * p.Cout + cv * 4;
                f32x4 o = o2[c] + b;
                if (p.resid) o = o + td_ld4(p.resid + off);
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
                } else if (p.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.01f * o[e];
                }
                td_st4(p.out + off, o);
            }
        }
    }
}

static inline int wino_tiles_1d(int n, int dil) { return ((n + dil - 1) / dil + 1) / 2; }
static inline long wino_tiles(int H, int W, int dil) { return (long)dil * dil * wino_tiles_1d(H, dil) * wino_tiles_1d(W, dil); }

// U = G g G^T (fp64) for every (co, ci): 16 [Cout][Cin] matrices, matrix xi*4+nu first
static inline void wino_transform_weights(const float* w, int Cout, int Cin, std::vector<std::vector<float>>& U) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    U.assign(16, std::vector<float>((size_t)Cout * Cin));
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* g = w + ((size_t)co * Cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    U[i * 4 + j][(size_t)co * Cin + ci] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}
