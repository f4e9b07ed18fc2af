// td_wino.h -- Winograd F(4x4, 3x3) for the stride-1 dilated 3x3 convolutions of layers 2-4 and the head (fp32).
//
// Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A  turns every m x m output tile into (m+2)^2 products per (ci, co) instead of
// 9 m^2: 36 instead of 144 for m = 4 (4x fewer MACs).  The contraction becomes 36 independent [tiles x Cin] x [Cin x Cout] GEMMs on the
// persistent fp32-MFMA GEMM (td_gemm.h / td_gemm_dma.h, nbatch = 36); the input/output transforms are HBM-bound passes over
// V = [36][T][Cin] and M = [36][T][Cout], 2.25x the size of the activation.  (F(2x2): 16 products instead of 36, 4x the activation in
// V / M -- rounds 1-4 carried it as tdnet_opts.winograd = 1 / 2; nothing routed to it since round 2 and it was removed in round 5,
// last commit 78dfa5a.  Its numerics are still on record: tests/numerics_winograd.py)
//
// A conv with dilation d (resnet.py:32-37: 2, 4, 8, 16 here) is d*d independent dilation-1 convs on the sub-grids
// {(py + d a, px + d b)}: a tile is (phase py, px; tile ty, tx) and its 4x4 input patch is read with stride d.
//   tile index t = ((py*d + px) * TY + ty) * TX + tx,  TY = ceil(ceil(H/d)/m), TX = ceil(ceil(W/d)/m)
// so all phases have the same tile count (out-of-range taps read zeros, out-of-range outputs are not written).
//
// Numerics: fp32 throughout (weights G g G^T are formed in fp64 on the host and rounded once).  Per conv the rms error vs fp64
// is 2.5x (F2) / 16x (F4, interpolation points 0, +-1, +-2) that of a direct fp32 conv; END TO END (td4 pipeline, CPU experiment
// tests/numerics_winograd.py) max|dlogit| vs fp64 is 1.2e-5 direct, 1.2e-5 F2, 2.6e-5 F4 -- BN, ReLU and the plane LayerNorm
// do not amplify it -- against a parity gate of 1e-3.
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
    // Rows per plane of V / M INCLUDING padding rows (>= T).  T*C*4 is a power of two for the frame's layers (4 MiB at 512 channels),
    // so 36 unpadded planes put the 36 concurrent streams of a transform on the same HBM channels; TP = T + pad de-phases them.
    int TP;
    // Optional plane-LayerNorm fused into the INPUT transform (the FCN head reads LayerNorm(feat), td4_psp18.py:151,306-312): the
    // patch element at pixel p, channel c becomes (x - ln_mean[c]) * ln_rstd[c] * ln_g[p] + ln_b[p] -- the arithmetic of k_ln_apply,
    // same operation order, so fused and unfused results are bit-identical -- and stays 0 outside the image (the conv's zero padding
    // applies to the normalised map).  ln_mean == nullptr: off.
    const float* ln_mean; const float* ln_rstd; const float* ln_g; const float* ln_b;
    // Chunked transforms (k_wino4_in_c / k_wino4_out_c): a launch covers the Tc tiles whose phase row is py = ny * i + cy and whose
    // phase column is px = nx * j + cx (i < dil / ny, j < dil / nx); V / Mb then hold ONLY those tiles ([36][TP][C], TP >= Tc).  With
    // an even dilation and ny = 2 the two chunks are the even and the odd image rows: a dilated conv maps a row parity onto itself,
    // so through a run of even-dilation convs the two chunks are independent chains (td_frame.h run_parity_chains).
    int Tc, ny, cy, nx, cx;
};
// Buffer descriptors of a transform: every access is an UNCONDITIONAL range-checked buffer access (a tap outside the image, an output
// pixel outside the map or a missing residual turn into an out-of-range offset / a zero-record descriptor) -- no branch around any
// load or store, so all loads of a thread are in flight together and the waits are exact counts.  With `if (inside) load` /
// `if (resid) load ... store` the residual of the output transform was one load, one wait and one store sixteen times over.
struct WinoBufs { TdBuf in, g, b, resid, out; };
TD_DEV WinoBufs td_wino_bufs(const WinoArgs& p) {
    WinoBufs w;
    const unsigned pix = (unsigned)p.H * (unsigned)p.W;
    w.in = td_make_buf(p.in, pix * (unsigned)p.C * 4u);
    w.g = td_make_buf(p.ln_g, p.ln_mean ? pix * 4u : 0u);
    w.b = td_make_buf(p.ln_b, p.ln_mean ? pix * 4u : 0u);
    w.resid = td_make_buf(p.resid, p.resid ? pix * (unsigned)p.Cout * 4u : 0u);
    w.out = td_make_buf(p.out, pix * (unsigned)p.Cout * 4u);
    return w;
}
// one patch element (4 channels); outside the image: zeros (also under the fused LayerNorm: gamma and beta read as 0 there)
TD_DEV f32x4 td_wino_ld(const WinoArgs& p, const WinoBufs& w, int y, int x, int cv, const f32x4& m4, const f32x4& r4) {
    const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const unsigned pix = (unsigned)y * (unsigned)p.W + (unsigned)x;
    f32x4 z = td_buf_ld4(w.in, ok ? (pix * (unsigned)p.C + (unsigned)cv * 4u) * 4u : TD_BUF_OOB, 0u);
    if (p.ln_mean) {                                                 // uniform
        const float g = td_buf_ld1(w.g, ok ? pix * 4u : TD_BUF_OOB, 0u), b = td_buf_ld1(w.b, ok ? pix * 4u : TD_BUF_OOB, 0u);
        z = (z - m4) * r4 * g + b;
    }
    return z;
}

// ---- F(4x4, 3x3): interpolation points 0, +-1, +-2, inf (Lavin & Gray) -----------------------------------------------------
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
TD_DEV void td_wino4_bt(const f32x4 (&d)[6], f32x4 (&t)[6]) {
    const f32x4 a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
TD_DEV void td_wino4_at(const f32x4 (&m)[6], f32x4 (&y)[4]) {
    const f32x4 p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = q + 2.f * s;
    y[2] = p + 4.f * r;
    y[3] = q + 8.f * s + m[5];
}

// thread = (tile, 4 channels): 6x6 patch (stride = dilation) -> 36 planes of V.  Columns first (a column is loaded, transformed
// and kept), then each row of the intermediate is transformed and stored: 144 VGPRs of live state instead of 288.
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_wino4_in(WinoArgs p) {
    const int CV = p.C >> 2;
    const WinoBufs wb = td_wino_bufs(p);
    const long total = (long)p.T * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int t = (int)(i / CV);
        const int tx = t % p.TX; t /= p.TX;
        const int ty = t % p.TY; t /= p.TY;
        const int px = t % p.dil, py = t / p.dil;
        f32x4 tm[6][6];
        f32x4 m4 = {0.f, 0.f, 0.f, 0.f}, r4 = m4;
        if (p.ln_mean) { m4 = td_ld4(p.ln_mean + cv * 4); r4 = td_ld4(p.ln_rstd + cv * 4); }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int x = px + p.dil * (4 * tx - 1 + c);
            f32x4 d[6], col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r] = td_wino_ld(p, wb, py + p.dil * (4 * ty - 1 + r), x, cv, m4, r4);
            td_wino4_bt(d, col);                                      // B^T d, one column
#pragma unroll
            for (int r = 0; r < 6; ++r) tm[r][c] = col[r];
        }
        const size_t tile = (size_t)(i / CV);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            f32x4 v[6];
            td_wino4_bt(tm[r], v);                                    // (.) B, one row
#pragma unroll
            for (int c = 0; c < 6; ++c) td_st4(p.V + ((size_t)(r * 6 + c) * p.TP + tile) * p.C + cv * 4, v[c]);
        }
    }
}

// thread = (tile, 4 output channels): Y = A^T m A (4x4 pixels), + bias (+ residual), activation, scatter
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_wino4_out(WinoArgs p) {
    const int CV = p.Cout >> 2;
    const WinoBufs wb = td_wino_bufs(p);
    const float slope = td_act_slope(p.act);
    const long total = (long)p.T * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const size_t tile = (size_t)(i / CV);
        int t = (int)tile;
        const int tx = t % p.TX; t /= p.TX;
        const int ty = t % p.TY; t /= p.TY;
        const int px = t % p.dil, py = t / p.dil;
        // the 16 residual vectors first: they are in flight under the 36 plane loads and the transforms (a residual load, a wait and
        // a store per output pixel made this kernel latency-bound: 32.5 us average against 25 for the larger input transform)
        unsigned offy[4], offx[4];
        bool oky[4], okx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = py + p.dil * (4 * ty + r), x = px + p.dil * (4 * tx + r);
            oky[r] = y < p.H; okx[r] = x < p.W;
            offy[r] = (unsigned)y * (unsigned)p.W * (unsigned)p.Cout * 4u;
            offx[r] = ((unsigned)x * (unsigned)p.Cout + (unsigned)cv * 4u) * 4u;
        }
        f32x4 rs[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) rs[r][c] = td_buf_ld4(wb.resid, (oky[r] && okx[c]) ? offy[r] + offx[c] : TD_BUF_OOB, 0u);
        f32x4 sm[4][6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            f32x4 m[6], col[4];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = td_ld4(p.Mb + ((size_t)(r * 6 + c) * p.TP + tile) * p.Cout + cv * 4);
            td_wino4_at(m, col);                                      // A^T m, one column
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[r][c] = col[r];
        }
        const f32x4 b = td_ld4(p.bias + cv * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 o4[4];
            td_wino4_at(sm[r], o4);                                   // (.) A, one row
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 o = o4[c] + b;
                o = o + rs[r][c];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = td_activate(o[e], slope);
                td_buf_st4(wb.out, (oky[r] && okx[c]) ? offy[r] + offx[c] : TD_BUF_OOB, o);
            }
        }
    }
}


// ---- low-register, chunk-aware F(4x4) transforms -----------------------------------------------------------------------------
// Same arithmetic as k_wino4_in / k_wino4_out, element for element (results are bit-identical), laid out for CO-RESIDENCY with the
// persistent GEMM: that kernel holds 3 workgroups per CU for its whole life at 136 VGPRs per wave, which leaves 104 registers
// per SIMD -- the float4-per-lane transforms above need 194 / 244 and could only start when a GEMM workgroup retires.  Here a
// WAVE owns (tile, slice of 64 * VW channels) and a lane VW channels (VW = 1: 60-odd VGPRs), so one transform wave fits beside
// three GEMM waves on every SIMD and the HBM-bound transform of one chunk runs UNDER the MFMA-bound GEMM of another.  The tile is
// wave-uniform: its decode (divisions by TX, TY, dil) and every pixel offset are scalar work, a lane adds its channel offset.
template <int VW> struct WinoVec;
template <> struct WinoVec<1> {
    typedef float T;
    TD_DEV_MEMBER T ld(TdBuf b, unsigned v, unsigned s) { return td_buf_ld1(b, v, s); }
    TD_DEV_MEMBER void st(TdBuf b, unsigned v, unsigned s, T x) { td_buf_st1(b, v, s, x); }
    TD_DEV_MEMBER T act(T x, float slope) { return td_activate(x, slope); }
};
template <> struct WinoVec<2> {
    typedef f32x2 T;
    TD_DEV_MEMBER T ld(TdBuf b, unsigned v, unsigned s) { return td_buf_ld2(b, v, s); }
    TD_DEV_MEMBER void st(TdBuf b, unsigned v, unsigned s, T x) { td_buf_st2(b, v, s, x); }
    TD_DEV_MEMBER T act(T x, float slope) { x[0] = td_activate(x[0], slope); x[1] = td_activate(x[1], slope); return x; }
};
template <> struct WinoVec<4> {
    typedef f32x4 T;
    // The wave-uniform part of the address is added into the per-lane offset: a 16-byte buffer STORE must not carry it in the SGPR
    // soffset (td_device.h td_buf_st4: the >64-bit store-data hazard the compiler does not see behind a register soffset).  The load
    // does the same -- not known to be needed, but it is the form that was verified on MI355X together with the store.
    TD_DEV_MEMBER T ld(TdBuf b, unsigned v, unsigned s) { return td_buf_ld4(b, v == TD_BUF_OOB ? v : v + s, 0u); }
    TD_DEV_MEMBER void st(TdBuf b, unsigned v, unsigned s, T x) { td_buf_st4(b, v == TD_BUF_OOB ? v : v + s, x); }
    TD_DEV_MEMBER T act(T x, float slope) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = td_activate(x[e], slope);
        return x;
    }
};
template <typename T>
TD_DEV void td_wino4_bt_t(const T (&d)[6], T (&t)[6]) {
    const T a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <typename T>
TD_DEV void td_wino4_at_t(const T (&m)[6], T (&y)[4]) {
    const T p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = q + 2.f * s;
    y[2] = p + 4.f * r;
    y[3] = q + 8.f * s + m[5];
}
// (wave-uniform) unit wv of a chunk -> tile, channel slice, phase and tile coordinates
struct WinoTile { int tl, sl, py, px, ty, tx; };
TD_DEV WinoTile td_wino_unit_tile(const WinoArgs& p, int slices, int wv) {
    WinoTile w;
    w.tl = wv / slices; w.sl = wv - w.tl * slices;
    int t = w.tl;
    w.tx = t % p.TX; t /= p.TX;
    w.ty = t % p.TY; t /= p.TY;
    const int pw = p.dil / p.nx;
    w.px = p.nx * (t % pw) + p.cx;
    w.py = p.ny * (t / pw) + p.cy;
    return w;
}

// One unit = one wave's work: (tile, slice of 64 VW channels) of the input transform.  `mid()` is called between the issue of the 36
// patch loads and their first use and again between the two 1-D passes: a no-op in the transform kernels (a hook the round-3 rider
// experiment used to park the wave on its workgroup's barrier with the loads in flight).
template <int VW, typename Mid>
TD_DEV void td_wino4_in_unit(const WinoArgs& p, int wv, Mid&& mid) {
    typedef WinoVec<VW> X;
    typedef typename X::T T;
    const int slices = (p.C + 64 * VW - 1) / (64 * VW);
    const WinoTile w = td_wino_unit_tile(p, slices, wv);
    const WinoBufs wb = td_wino_bufs(p);
    const int c0 = (w.sl * 64 + (int)(threadIdx.x & 63)) * VW;        // this lane's first channel
    const unsigned coff = c0 < p.C ? (unsigned)c0 * 4u : TD_BUF_OOB;  // lanes past C (C not a multiple of 64 VW): nothing read, nothing written
    T m4 = T(0.f), r4 = T(0.f);
    if (p.ln_mean) {
        const TdBuf mb = td_make_buf(p.ln_mean, (unsigned)p.C * 4u), rb = td_make_buf(p.ln_rstd, (unsigned)p.C * 4u);
        m4 = X::ld(mb, coff, 0u); r4 = X::ld(rb, coff, 0u);
    }
    T dd[6][6];                                                       // [c][r]
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int x = w.px + p.dil * (4 * w.tx - 1 + c);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int y = w.py + p.dil * (4 * w.ty - 1 + r);
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;          // wave-uniform
            const unsigned pix = (unsigned)y * (unsigned)p.W + (unsigned)x;
            T z = X::ld(wb.in, ok ? coff : TD_BUF_OOB, ok ? pix * (unsigned)p.C * 4u : 0u);
            if (p.ln_mean) {                                         // uniform; the arithmetic of td_wino_ld / k_ln_apply, same order (the head conv only)
                const float g = td_buf_ld1(wb.g, ok ? 0u : TD_BUF_OOB, ok ? pix * 4u : 0u), b = td_buf_ld1(wb.b, ok ? 0u : TD_BUF_OOB, ok ? pix * 4u : 0u);
                z = (z - m4) * r4 * g + b;
            }
            dd[c][r] = z;
        }
    }
    mid();
    T tm[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        T col[6];
        td_wino4_bt_t(dd[c], col);                                    // B^T d, one column
#pragma unroll
        for (int r = 0; r < 6; ++r) tm[r][c] = col[r];
    }
    mid();
    const unsigned plane = (unsigned)p.TP * (unsigned)p.C * 4u;
    const TdBuf vb = td_make_buf(p.V, 36u * plane);
    const unsigned voff = (unsigned)w.tl * (unsigned)p.C * 4u;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        T v[6];
        td_wino4_bt_t(tm[r], v);                                      // (.) B, one row
#pragma unroll
        for (int c = 0; c < 6; ++c) X::st(vb, coff, (unsigned)(r * 6 + c) * plane + voff, v[c]);
    }
}

template <int VW, typename Mid>
TD_DEV void td_wino4_out_unit(const WinoArgs& p, int wv, Mid&& mid) {
    typedef WinoVec<VW> X;
    typedef typename X::T T;
    const int slices = (p.Cout + 64 * VW - 1) / (64 * VW);
    const WinoTile w = td_wino_unit_tile(p, slices, wv);
    const WinoBufs wb = td_wino_bufs(p);
    const float slope = td_act_slope(p.act);
    const int c0 = (w.sl * 64 + (int)(threadIdx.x & 63)) * VW;
    const unsigned coff = c0 < p.Cout ? (unsigned)c0 * 4u : TD_BUF_OOB;
    unsigned offy[4], offx[4];
    bool oky[4], okx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = w.py + p.dil * (4 * w.ty + r), x = w.px + p.dil * (4 * w.tx + r);
        oky[r] = y < p.H; okx[r] = x < p.W;
        offy[r] = (unsigned)y * (unsigned)p.W * (unsigned)p.Cout * 4u;
        offx[r] = (unsigned)x * (unsigned)p.Cout * 4u;
    }
    T rs[4][4];                                                       // residual first: in flight under the 36 plane loads
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = oky[r] && okx[c];
            rs[r][c] = X::ld(wb.resid, ok ? coff : TD_BUF_OOB, ok ? offy[r] + offx[c] : 0u);
        }
    const unsigned plane = (unsigned)p.TP * (unsigned)p.Cout * 4u;
    const TdBuf mb = td_make_buf(p.Mb, 36u * plane);
    const unsigned moff = (unsigned)w.tl * (unsigned)p.Cout * 4u;
    T mm[6][6];                                                       // [c][r]
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = 0; r < 6; ++r) mm[c][r] = X::ld(mb, coff, (unsigned)(r * 6 + c) * plane + moff);
    const TdBuf bbuf = td_make_buf(p.bias, (unsigned)p.Cout * 4u);
    const T b = X::ld(bbuf, coff, 0u);
    mid();
    T sm[4][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        T col[4];
        td_wino4_at_t(mm[c], col);                                    // A^T m, one column
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[r][c] = col[r];
    }
    mid();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T o4[4];
        td_wino4_at_t(sm[r], o4);                                     // (.) A, one row
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = oky[r] && okx[c];
            T o = o4[c] + b;
            o = o + rs[r][c];
            X::st(wb.out, ok ? coff : TD_BUF_OOB, ok ? offy[r] + offx[c] : 0u, X::act(o, slope));
        }
    }
}

template <int VW>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, VW == 4 ? 2 : 4) k_wino4_in_c(WinoArgs p) {
    const int slices = (p.C + 64 * VW - 1) / (64 * VW);
    const int wv = TD_UNIFORM((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wv >= p.Tc * slices) return;
    td_wino4_in_unit<VW>(p, wv, []() {});
}
template <int VW>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, VW == 4 ? 2 : 4) k_wino4_out_c(WinoArgs p) {
    const int slices = (p.Cout + 64 * VW - 1) / (64 * VW);
    const int wv = TD_UNIFORM((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wv >= p.Tc * slices) return;
    td_wino4_out_unit<VW>(p, wv, []() {});
}
// ---- the FCN head's output transform + its 1x1 classifier in one launch (round 6; td4_psp18.py:295-299: conv3x3 -> BN -> ReLU -> conv1x1 + bias) -------------
// A wave owns one tile = 16 pixels x ALL Cout = 64 VW channels of the head's hidden map (td4: 128, td2: 64): A^T m A + bias, ReLU as in td_wino4_out_unit, the
// 16 x Cout values go to LDS (rows padded by 4 floats: 16 lanes reading 16 rows hit 16 different 16-byte slots) instead of HBM, and lane i computes the outputs
// (pixel i & 15, class i >> 4), (.., + 4), .. with k_classifier's summation order -- four sequential fma chains over the channel quarters, added as
// ((s0 + s1) + s2) + s3 + bias -- so the low-resolution logits are bit-identical to the two-kernel form.  The hidden map (17 MB at 1024x2048) is never written.
struct ClsArgs { const float* w; const float* b; float* out; int NC; };   // classifier [NC][Cout], [NC]; out planar [NC][H * W]
template <int VW>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_wino4_out_cls(WinoArgs p, ClsArgs c) {
    typedef WinoVec<VW> X;
    typedef typename X::T T;
    constexpr int C = 64 * VW, ROW = C + 4;
    TD_DYN_LDS(smem);
    float* ws = reinterpret_cast<float*>(smem);                        // [NC][C]
    float* ys = ws + c.NC * C + (threadIdx.x >> 6) * (16 * ROW);       // this wave's [16 pixels][ROW]
    for (int i = threadIdx.x; i < c.NC * C; i += blockDim.x) ws[i] = c.w[i];
    __syncthreads();                                                   // the only workgroup barrier: waves past the last tile may leave after it
    const int wv = TD_UNIFORM((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wv >= p.Tc) return;
    const int lane = threadIdx.x & 63;
    const WinoTile w = td_wino_unit_tile(p, 1, wv);
    const float slope = td_act_slope(p.act);
    const unsigned coff = (unsigned)(lane * VW) * 4u;
    const unsigned plane = (unsigned)p.TP * (unsigned)C * 4u;
    const TdBuf mb = td_make_buf(p.Mb, 36u * plane);
    const unsigned moff = (unsigned)w.tl * (unsigned)C * 4u;
    T mm[6][6];                                                       // [c][r]
#pragma unroll
    for (int cc = 0; cc < 6; ++cc)
#pragma unroll
        for (int r = 0; r < 6; ++r) mm[cc][r] = X::ld(mb, coff, (unsigned)(r * 6 + cc) * plane + moff);
    const TdBuf bbuf = td_make_buf(p.bias, (unsigned)C * 4u);
    const T b = X::ld(bbuf, coff, 0u);
    T sm[4][6];
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
        T col[4];
        td_wino4_at_t(mm[cc], col);                                   // A^T m, one column
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[r][cc] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T o4[4];
        td_wino4_at_t(sm[r], o4);                                     // (.) A, one row
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            T o = o4[cc] + b;
            o = o + T(0.f);                                           // the zero residual of td_wino4_out_unit / k_wino4_out (-0 + 0 = +0: same bits)
            o = X::act(o, slope);
            float* dst = ys + (r * 4 + cc) * ROW + lane * VW;
            if constexpr (VW == 1) dst[0] = o;
            else { dst[0] = o[0]; dst[1] = o[1]; }
        }
    }
    td_wave_sync();
    const int CQ = C >> 2;
    for (int idx = lane; idx < 16 * c.NC; idx += 64) {
        const int px = idx & 15, k = idx >> 4;
        const float* yr = ys + px * ROW;
        const float* wr = ws + k * C;
        float s[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float a = 0.f;
            for (int ch = 0; ch < CQ; ch += 4) {
                const f32x4 v = td_ld4(yr + qd * CQ + ch), u = td_ld4(wr + qd * CQ + ch);
                a = fmaf(v[0], u[0], a); a = fmaf(v[1], u[1], a); a = fmaf(v[2], u[2], a); a = fmaf(v[3], u[3], a);
            }
            s[qd] = a;
        }
        const int y = w.py + p.dil * (4 * w.ty + (px >> 2)), x = w.px + p.dil * (4 * w.tx + (px & 3));
        if (y < p.H && x < p.W) c.out[(size_t)k * p.H * p.W + (size_t)y * p.W + x] = (((s[0] + s[1]) + s[2]) + s[3]) + c.b[k];
    }
}
static inline bool wino_out_cls_supports(int Cout, int NC) { return (Cout == 64 || Cout == 128) && NC >= 1 && NC <= 32; }
static inline int wino_out_cls_lds(int Cout, int NC) { return (NC * Cout + 4 * 16 * (Cout + 4)) * 4; }

// grid of a chunked transform: one wave per (tile, channel slice)
static inline unsigned wino_chunk_grid(int Tc, int C, int VW) { return (unsigned)(((long)Tc * ((C + 64 * VW - 1) / (64 * VW)) + 3) / 4); }


// m = output tile edge (2 or 4)
static inline int wino_tiles_1d(int n, int dil, int m) { return ((n + dil - 1) / dil + m - 1) / m; }
static inline long wino_tiles(int H, int W, int dil, int m) { return (long)dil * dil * wino_tiles_1d(H, dil, m) * wino_tiles_1d(W, dil, m); }

// tile count for an output of M pixels whose shape is not at hand (layer setup): assume the 1:2 frames of the path
static inline long wino_tiles_estimate(long M, int dil, int m) {
    int H = 1;
    while ((long)H * H * 2 < M) ++H;
    const int W = (int)((M + H - 1) / H);
    return wino_tiles(H, W, dil, m);
}

// U = G g G^T (fp64) for every (co, ci): (m+2)^2 [Cout][Cin] matrices, matrix xi*(m+2)+nu first
static inline void wino_transform_weights(const float* w, int Cout, int Cin, int m, std::vector<std::vector<float>>& U) {
    static const double G4[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                    {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int n = m + 2;
    const double (*G)[3] = G4;                                        // m == 4 only
    U.assign((size_t)n * n, std::vector<float>((size_t)Cout * Cin));
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* g = w + ((size_t)co * Cin + ci) * 9;
            double t[6][3];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    U[(size_t)i * n + j][(size_t)co * Cin + ci] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}
