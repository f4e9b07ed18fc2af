// td_attn_h.h -- the attention-propagation kernel of td_attn.h on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate):
// BASELINE.json config 5 ("fp16 MFMA"), tdnet_opts.precision = 1.       out = softmax(q k^T / sqrt(64)) v' + bias (+ resid)
//
// Interfaces stay fp32 (q [Lq][64], k [Lk][64], bias, resid, out [Lq][DV], cache entries): operands are rounded to fp16 on the way
// into the MFMAs, the softmax (scores, running reference, exp2, row sums) is fp32, P is rounded to fp16 for the P V' product.
// 16x the matrix rate of the fp32 kernel: 4 + 8 NT MFMAs of 32 cycles per 32 x (32 CW) key tile instead of 32 + 64 NT of 64.
//
// Differences from td_attn.h forced by the operand format (a lane supplies 8 CONSECUTIVE k of its row / column):
//   * V' must be read along the key axis (8 consecutive keys of one channel per lane), so it is consumed re-tiled and already in
//     fp16: vt [LkPad / 8][DV][8 halfs] -- key group major, so the 32 lanes of a channel tile read 32 consecutive 16-byte pieces
//     (k_attn_vt_h below, a 2 MB pass; LkPad = Lk rounded up to the key super-tile, zero filled: no V' load is ever ragged).  (A
//     plain transpose [DV][LkPad] was measured first: every lane then reads from its own 4 KB-strided row, 64 cache lines per load
//     instruction, and the kernel ran at fp32 speed.)
//   * P goes through LDS in the natural [key group of 8][query][8 halfs] image (conflict-free 16-byte reads), written by the
//     lanes that own the scores as four 8-byte pieces;
//   * the accumulator of channel tile j holds channel cb0 + 32 j + (lane & 31) (not NT consecutive channels per lane): a wave's
//     store of one output row is 128 contiguous bytes per tile.
// Softmax schedule: the single-pass one of td_attn.h (per-query reference moved only when a tile exceeds it by more than TAU).
#pragma once
#include "td_attn.h"
#include "td_conv_h.h"   // f16x8 helpers

struct AttnArgsH {
    const float* q;          // [Lq][64]
    const float* k;          // [Lk][64]
    const _Float16* vt;      // [LkPad / 8][ldv][8]: V' in key groups of 8, fp16, zero beyond Lk (pointer already at the launch's first channel)
    const float* bias;       // [DV] or nullptr
    const float* resid;      // [Lq][ldv] or nullptr
    float* out;              // [Lq][ldv]
    int Lq, Lk, LkPad;
    float scale_log2e;
    float* ln_part;          // optional plane-LayerNorm strip statistics of out (td_attn.h)
    int ln_nstr;
    int ldv;                 // row stride of resid / out / ln_part (floats)
};

// vp [Lk][ldv] fp32 -> vt [LkPad / 8][ldv][8] fp16, zero padded.  grid = (LkPad/64, DV/64), block 256: a 64 x 64 tile through LDS
// so that both the reads (along channels) and the writes (16 bytes per channel, channels consecutive) are contiguous.
TD_KERNEL void k_attn_vt_h(const float* __restrict__ vp, _Float16* __restrict__ vt, int Lk, int LkPad, int ldv) {
    TD_DYN_LDS(smem);
    float* tile = reinterpret_cast<float*>(smem);                 // [64 keys][65]
    const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int kk = i >> 6, cc = i & 63;
        tile[kk * 65 + cc] = (k0 + kk < Lk) ? vp[(size_t)(k0 + kk) * ldv + c0 + cc] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int e = i & 7, cc = (i >> 3) & 63, kg = i >> 9;
        if (k0 + 8 * kg < LkPad) vt[((size_t)(k0 / 8 + kg) * ldv + c0 + cc) * 8 + e] = (_Float16)tile[(8 * kg + e) * 65 + cc];
    }
}

template <int QW, int CW>
struct AttnLdsH {
    static constexpr int P_HALFS = QW * (4 * CW) * 32 * 8;       // one super-tile of P: [qw][key group of 8][q][8 halfs]
    static constexpr int RED_FLOATS = QW * CW * 32;
    static constexpr int BYTES = 2 * P_HALFS * 2 + 3 * RED_FLOATS * 4;
};

template <int QW, int CW, int NT>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_attention_h(AttnArgsH p) {
    static_assert(QW * CW == 4, "4 waves per block");
    constexpr int DV = CW * NT * 32;
    constexpr int SK = 32 * CW;                                  // keys per super-tile
    using L = AttnLdsH<QW, CW>;
    TD_DYN_LDS(smem);
    _Float16* Ps = reinterpret_cast<_Float16*>(smem);            // [2][P_HALFS]
    float* red = reinterpret_cast<float*>(Ps + 2 * L::P_HALFS);  // [QW][CW][32] tile-max exchange
    float* red2 = red + L::RED_FLOATS;                           // row-sum exchange
    float* scr = red2 + L::RED_FLOATS;                           // [4 waves][32] rescale factors

    const int tid = threadIdx.x, lane = tid & 63, wave = TD_UNIFORM(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int qw = wave / CW, cw = wave % CW;
    const int q0 = (blockIdx.x * QW + qw) * 32;
    const unsigned LDV = (unsigned)p.ldv;

    // this lane's query row (column l31 of Q^T), pre-scaled, as the four k16-step fragments: d = 16 ks + 8 half + (0..7)
    f16x8 qf[4];
    {
        const int q = (q0 + l31 < p.Lq) ? q0 + l31 : p.Lq - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* src = p.q + (size_t)q * 64 + 16 * ks + 8 * half;
            qf[ks] = td_cvt8(td_ld4(src) * p.scale_log2e, td_ld4(src + 4) * p.scale_log2e);
        }
    }
    const int key_last = p.Lk - 1;
    auto load_k = [&](int kb, f32x4 (&kf)[8]) {                  // row (key) kb + l31, same d ranges; clamped past the end (masked below)
        const int key = (kb + l31 < p.Lk) ? kb + l31 : key_last;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* src = p.k + (size_t)key * 64 + 16 * ks + 8 * half;
            kf[2 * ks] = td_ld4(src);
            kf[2 * ks + 1] = td_ld4(src + 4);
        }
    };
    auto score_tile = [&](const f32x4 (&kf)[8]) -> f32x16 {      // S^T tile: this lane's query against keys (r&3) + 8 (r>>2) + 4 half
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = td_mfma32_f16(td_cvt8(kf[2 * ks], kf[2 * ks + 1]), qf[ks], s);
        return s;
    };

    const int nsuper = (p.Lk + SK - 1) / SK;
    const float NEG = -3.0e38f;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float lsum = 0.f, rowmax = NEG;
    const int cb0 = cw * (NT * 32);                               // first channel of this wave; tile j, lane l31 -> channel cb0 + 32 j + l31
    const _Float16* vch[NT];                                      // key group `half` of a k16-step, this lane's channel of tile j
#pragma unroll
    for (int j = 0; j < NT; ++j) vch[j] = p.vt + ((size_t)half * LDV + cb0 + 32 * j + l31) * 8;
    const size_t vgroup = (size_t)LDV * 8;                        // halfs per key group

    f32x4 kf[8];
    load_k(cw * 32, kf);
    for (int st = 0; st < nsuper; ++st) {
        const int kbase = st * SK, kb = kbase + cw * 32;
        const f32x16 s = score_tile(kf);
        float lm = NEG;
        if (kb + 32 <= p.Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lm = __builtin_fmaxf(lm, s[r]);   // v_max(3)_f32: half the instructions of compare + select
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
                lm = (key < p.Lk && s[r] > lm) ? s[r] : lm;
            }
        }
        lm = fmaxf(lm, td_shfl_xor(lm, 32));
        if (half == 0) red[(qw * CW + cw) * 32 + l31] = lm;
        __syncthreads();
        if (st + 1 < nsuper) load_k(kb + SK, kf);                 // next key tile, in flight under the P V' MFMAs
        float tm = NEG;
#pragma unroll
        for (int c = 0; c < CW; ++c) tm = fmaxf(tm, red[(qw * CW + c) * 32 + l31]);
        if (td_any(tm > rowmax + 8.0f)) {                         // identical decision in the CW waves of a query tile (same maxima, same references)
            const float nm = fmaxf(rowmax, tm);
            const float alpha = td_exp2(rowmax - nm);
            lsum *= alpha;
            rowmax = nm;
            float* sc = scr + wave * 32;
            if (half == 0) sc[l31] = alpha;
            td_wave_sync();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 a4 = td_ld4(sc + 8 * u + 4 * half);   // accumulator rows 8u + 4 half + {0..3} = registers 4u + {0..3}
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j][4 * u + e] *= a4[e];
            }
            td_wave_sync();
        }
        // P = exp2(S - reference) -> fp16 -> LDS image [key group][query][8]: this lane's registers 4g .. 4g+3 are keys 8g + 4 half + (0..3)
        _Float16* Pw = Ps + (st & 1) * L::P_HALFS + qw * (4 * CW * 256);
        f32x16 pr;
        if (kb + 32 <= p.Lk) {                                        // wave-uniform: no key of this tile is masked
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = td_exp2(s[r] - rowmax);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = (kb + 8 * (r >> 2) + 4 * half + (r & 3) < p.Lk) ? td_exp2(s[r] - rowmax) : 0.f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 ph = (_Float16)pr[4 * g + e];
                h4[e] = ph;
                lsum += (float)ph;                                // the row sum of what is actually multiplied (fp16-rounded P)
            }
            *reinterpret_cast<f16x4*>(Pw + ((cw * 4 + g) * 32 + l31) * 8 + 4 * half) = h4;
        }
        __syncthreads();
        // O += P V': k16-step ks covers keys kbase + 16 ks + 8 half + (0..7); A = P image group 2 ks + half, B = 16 bytes of vt
#pragma unroll
        for (int ks = 0; ks < 2 * CW; ++ks) {
            const f16x8 a8 = td_ld8h(Pw + ((2 * ks + half) * 32 + l31) * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = td_mfma32_f16(a8, td_ld8h(vch[j] + (size_t)(kbase / 8 + 2 * ks) * vgroup), acc[j]);
        }
    }
    // ---- row sums -> 1/l, epilogue ---------------------------------------------------------------------------
    lsum += td_shfl_xor(lsum, 32);
    if (half == 0) red2[(qw * CW + cw) * 32 + l31] = lsum;
    __syncthreads();
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = p.bias ? p.bias[cb0 + 32 * j + l31] : 0.f;
    const bool ln = p.ln_part != nullptr;
    float kshift[NT], s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) kshift[j] = s1[j] = s2[j] = 0.f;
    if (ln && q0 < p.Lq) {                                        // shift = the strip's first row (register 0 of lane-half 0), see td_attn.h
        float l = 0.f;
#pragma unroll
        for (int c = 0; c < CW; ++c) l += red2[(qw * CW + c) * 32];
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = acc[j][0] * inv + bv[j];
            if (p.resid) v += p.resid[(size_t)q0 * LDV + cb0 + 32 * j + l31];
            const float other = td_shfl_xor(v, 32);
            kshift[j] = half ? other : v;
        }
    }
    // buffer-addressed (32-bit offsets; rows >= Lq fall outside the descriptor), residual rows requested four at a time before the first
    // is used (td_attn.h); no residual: a descriptor with zero records
    const TdBuf out_buf = td_make_buf(p.out, 0x80000000u);
    const TdBuf res_buf = td_make_buf(p.resid, p.resid ? 0x80000000u : 0u);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        float rv[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rg * 4 + i;
            const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const unsigned off = q < p.Lq ? ((unsigned)q * (unsigned)LDV + (unsigned)(cb0 + l31)) * 4u : TD_BUF_OOB;
#pragma unroll
            for (int j = 0; j < NT; ++j) rv[i][j] = td_buf_ld1(res_buf, off, (unsigned)(128 * j));
        }
        TD_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rg * 4 + i;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int q = q0 + row;
            const bool live = q < p.Lq;
            const unsigned off = live ? ((unsigned)q * (unsigned)LDV + (unsigned)(cb0 + l31)) * 4u : TD_BUF_OOB;
            float l = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) l += red2[(qw * CW + c) * 32 + row];
            const float inv = 1.0f / l;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float o = acc[j][r] * inv + bv[j] + rv[i][j];
                td_buf_st1(out_buf, off, (unsigned)(128 * j), o);
                if (ln && live) { const float d = o - kshift[j]; s1[j] += d; s2[j] += d * d; }
            }
        }
    }
    if (ln) {
        const int cnt = p.Lq - q0 < 32 ? (p.Lq - q0 > 0 ? p.Lq - q0 : 0) : 32;
        const int strip = blockIdx.x * QW + qw;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float t1 = s1[j] + td_shfl_xor(s1[j], 32), t2 = s2[j] + td_shfl_xor(s2[j], 32);
            float mean = 0.f, m2 = 0.f;
            if (cnt) {
                const float rn = 1.0f / (float)cnt;
                mean = kshift[j] + t1 * rn;
                const float v = t2 - t1 * t1 * rn;
                m2 = v > 0.f ? v : 0.f;
            }
            if (half == 0) {
                p.ln_part[(size_t)strip * LDV + cb0 + 32 * j + l31] = mean;
                p.ln_part[((size_t)p.ln_nstr + strip) * LDV + cb0 + 32 * j + l31] = m2;
            }
        }
    }
}

static inline int attn_lkpad(int Lk) { return (Lk + 127) / 128 * 128; }

// V' [Lk][DV] fp32 -> the re-tiled fp16 operand of k_attention_h.  A launch of its own so that the caller can put it where V' is PRODUCED
// (the side stream, under the backbone) instead of in front of the attention that consumes it (td_frame.h launch_chain).
static inline void attn_prepare_vt_h(const float* vp, int Lk, int DV, _Float16* vt, hipStream_t s) {
    const int LkPad = attn_lkpad(Lk);
    TD_LAUNCH(k_attn_vt_h, dim3(LkPad / 64, DV / 64), dim3(256), 64 * 65 * 4, s, vp, vt, Lk, LkPad, DV);
}
// vt: workspace of DV * attn_lkpad(Lk) halfs (the caller's; filled here from vp on the same stream unless vt_ready says it already holds a.vp)
static inline int attn_launch_h(const AttnArgs& a, int DV, _Float16* vt, hipStream_t s, bool vt_ready = false) {
    const int LkPad = attn_lkpad(a.Lk);
    if (DV != 128 && (DV < 512 || DV % 512)) return -1;
    if (!vt_ready) attn_prepare_vt_h(a.vp, a.Lk, DV, vt, s);
    AttnArgsH h;
    h.q = a.q; h.k = a.k; h.vt = vt; h.bias = a.bias; h.resid = a.resid; h.out = a.out; h.Lq = a.Lq; h.Lk = a.Lk; h.LkPad = LkPad;
    h.scale_log2e = a.scale_log2e; h.ln_part = a.ln_part; h.ldv = DV;
    if (DV == 128) {
        const int grid = (a.Lq + 63) / 64;
        h.ln_nstr = 2 * grid;
        TD_LAUNCH((k_attention_h<2, 2, 2>), dim3(grid), dim3(256), (AttnLdsH<2, 2>::BYTES), s, h);
        return 0;
    }
    const int grid = (a.Lq + 31) / 32;
    h.ln_nstr = grid;
    for (int c0 = 0; c0 < DV; c0 += 512) {
        AttnArgsH b = h;
        b.vt += (size_t)c0 * 8; b.out += c0;
        if (b.bias) b.bias += c0;
        if (b.resid) b.resid += c0;
        if (b.ln_part) b.ln_part += c0;
        TD_LAUNCH((k_attention_h<1, 4, 4>), dim3(grid), dim3(256), (AttnLdsH<1, 4>::BYTES), s, b);
    }
    return 0;
}
