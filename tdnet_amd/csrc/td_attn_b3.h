// td_attn_b3.h -- the attention-propagation kernel of td_attn.h with fp32-ACCURATE products on the bf16 MFMA (tdnet_opts.precision = 2):
//                                     out = softmax(q k^T / sqrt(64)) v' + bias (+ resid)
// Every fp32 operand of the two contractions (q, k, P = exp2(S - reference), v') is the exact sum of three bf16 parts (td_gemm_b3.h) and every
// product is six bf16-MFMA products with fp32 accumulation: S and P V' carry fp32 accuracy at 16 / 6 the fp32 MFMA's rate.  The softmax itself
// (scores, running reference, exp2, row sums) is the fp32 arithmetic of td_attn.h's single-pass schedule; the row sum is the sum of the fp32 P (its
// three parts add up to it exactly).
//
// Structure: the fp16 kernel's (td_attn_h.h) -- a lane supplies 8 CONSECUTIVE k of its row / column to v_mfma_f32_32x32x16_bf16 --
//   * q: split once per query tile (4 k16-steps x 3 parts); k: read as fp32 from the cache entry and split per key tile (176 VALU per 24 MFMAs of
//     the score tile); P: split by the lanes that own the scores and published through LDS as three [key group of 8][query][8 bf16] images;
//   * v' is consumed re-tiled AND pre-split: vt3 [part 3][LkPad / 8][DV][8 bf16] (k_attn_vt_b3: one 6 MB pass at Lk = 2048, on the side stream
//     where V' is produced), so a lane's B operand of a part is one 16-byte load and 32 lanes of a channel tile read 512 contiguous bytes.
// (Round 6 also ran a second form -- the query parts kept in LDS instead of 48 registers, v' fragments by buffer loads with SGPR offsets, every k16-step's
// fragments requested one step ahead and the first step's before the barrier that publishes P: 196 instead of 242 VGPRs, and 1.5-2.6 % SLOWER in the frame,
// interleaved in one process (311.7 -> 307.0 frames/s at 1024x2048, 429.6 -> 418.7 at 769x1537; with a bare barrier 322.4 -> 315.1: profiles/r06n_*).  With two
// workgroups per CU the compiler's just-in-time loads interleave better than the bulk prefetch.  Removed.)
#pragma once
#include "td_attn_h.h"
#include "td_gemm_b3.h"

struct AttnArgsB3 {
    const float* q;          // [Lq][64]
    const float* k;          // [Lk][64]
    const unsigned short* vt;   // [3][LkPad / 8][ldv][8] bf16 parts of V' in key groups of 8, zero beyond Lk (pointer already at the launch's first channel)
    const float* bias;       // [DV] or nullptr
    const float* resid;      // [Lq][ldv] or nullptr
    float* out;              // [Lq][ldv]
    int Lq, Lk, LkPad;
    float scale_log2e;
    float* ln_part;          // optional plane-LayerNorm strip statistics of out (td_attn.h)
    int ln_nstr;
    int ldv;
    size_t part_stride;      // elements between two parts of vt: (LkPad / 8) * full DV * 8
};

// vp [Lk][ldv] fp32 -> vt3 [3][LkPad / 8][ldv][8] bf16 parts, zero padded.  grid = (LkPad/64, DV/64), block 256: a 64 x 64 tile through LDS
TD_KERNEL void k_attn_vt_b3(const float* __restrict__ vp, unsigned short* __restrict__ vt, int Lk, int LkPad, int ldv) {
    TD_DYN_LDS(smem);
    float* tile = reinterpret_cast<float*>(smem);                 // [64 keys][65]
    const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int kk = i >> 6, cc = i & 63;
        tile[kk * 65 + cc] = (k0 + kk < Lk) ? vp[(size_t)(k0 + kk) * ldv + c0 + cc] : 0.f;
    }
    __syncthreads();
    const size_t part = (size_t)(LkPad / 8) * ldv * 8;
    for (int i = threadIdx.x; i < 64 * 64 / 2; i += blockDim.x) {   // a pair of consecutive keys of one channel per thread
        const int e = (i & 3) * 2, cc = (i >> 2) & 63, kg = i >> 8;
        if (k0 + 8 * kg >= LkPad) continue;
        unsigned h, m, l;
        td_split3_pair(tile[(8 * kg + e) * 65 + cc], tile[(8 * kg + e + 1) * 65 + cc], h, m, l);
        const size_t o = ((size_t)(k0 / 8 + kg) * ldv + c0 + cc) * 8 + e;
        *reinterpret_cast<unsigned*>(vt + o) = h;
        *reinterpret_cast<unsigned*>(vt + part + o) = m;
        *reinterpret_cast<unsigned*>(vt + 2 * part + o) = l;
    }
}

template <int QW, int CW>
struct AttnLdsB3 {
    static constexpr int P_BYTES = QW * (4 * CW) * 32 * 16;      // one part of one super-tile of P: [qw][key group of 8][q][8 bf16]
    static constexpr int RED_FLOATS = QW * CW * 32;
    static constexpr int BYTES = 2 * 3 * P_BYTES + 3 * RED_FLOATS * 4;
};

template <int QW, int CW, int NT>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_attention_b3(AttnArgsB3 p) {
    static_assert(QW * CW == 4, "4 waves per block");
    constexpr int SK = 32 * CW;                                  // keys per super-tile
    using L = AttnLdsB3<QW, CW>;
    TD_DYN_LDS(smem);
    char* Ps = smem;                                             // [2 buffers][3 parts][P_BYTES]
    float* red = reinterpret_cast<float*>(Ps + 2 * 3 * L::P_BYTES);   // [QW][CW][32] tile-max exchange
    float* red2 = red + L::RED_FLOATS;                           // row-sum exchange
    float* scr = red2 + L::RED_FLOATS;                           // [4 waves][32] rescale factors

    const int tid = threadIdx.x, lane = tid & 63, wave = TD_UNIFORM(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int qw = wave / CW, cw = wave % CW;
    const int q0 = (blockIdx.x * QW + qw) * 32;
    const unsigned LDV = (unsigned)p.ldv;

    // this lane's query row (column l31 of Q^T), pre-scaled, as the four k16-step fragments x three parts: d = 16 ks + 8 half + (0..7)
    u32x4 qh[4], qm[4], ql[4];
    {
        const int q = (q0 + l31 < p.Lq) ? q0 + l31 : p.Lq - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* src = p.q + (size_t)q * 64 + 16 * ks + 8 * half;
            td_split3(td_ld4(src) * p.scale_log2e, td_ld4(src + 4) * p.scale_log2e, qh[ks], qm[ks], ql[ks]);
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
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 kh, km, kl;
            td_split3(kf[2 * ks], kf[2 * ks + 1], kh, km, kl);
            s = td_mfma32_bf16(kh, qh[ks], s);
            s = td_mfma32_bf16(kh, qm[ks], s);
            s = td_mfma32_bf16(km, qh[ks], s);
            s = td_mfma32_bf16(kh, ql[ks], s);
            s = td_mfma32_bf16(km, qm[ks], s);
            s = td_mfma32_bf16(kl, qh[ks], s);
        }
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
    const unsigned short* vch[NT];                                // key group `half` of a k16-step, this lane's channel of tile j (part 0)
#pragma unroll
    for (int j = 0; j < NT; ++j) vch[j] = p.vt + ((size_t)half * LDV + cb0 + 32 * j + l31) * 8;
    const size_t vgroup = (size_t)LDV * 8;                        // elements per key group
    const size_t vpart = p.part_stride;

    f32x4 kf[8];
    load_k(cw * 32, kf);
    for (int st = 0; st < nsuper; ++st) {
        const int kbase = st * SK, kb = kbase + cw * 32;
        const f32x16 s = score_tile(kf);
        float lm = NEG;
        if (kb + 32 <= p.Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lm = __builtin_fmaxf(lm, s[r]);
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
        // P = exp2(S - reference), fp32 -> three bf16 parts -> LDS images [part][key group][query][8]: this lane's registers 4g .. 4g+3 are
        // keys 8g + 4 half + (0..3): two dwords of each part
        char* Pw = Ps + (st & 1) * (3 * L::P_BYTES) + qw * (4 * CW * 512);
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
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 h2, m2, l2;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                unsigned h_, m_, l_;
                td_split3_pair(pr[4 * g + 2 * e2], pr[4 * g + 2 * e2 + 1], h_, m_, l_);
                h2[e2] = h_; m2[e2] = m_; l2[e2] = l_;
                lsum += pr[4 * g + 2 * e2] + pr[4 * g + 2 * e2 + 1];
            }
            const int off = ((cw * 4 + g) * 32 + l31) * 16 + 8 * half;
            *reinterpret_cast<u32x2*>(Pw + off) = h2;
            *reinterpret_cast<u32x2*>(Pw + L::P_BYTES + off) = m2;
            *reinterpret_cast<u32x2*>(Pw + 2 * L::P_BYTES + off) = l2;
        }
        __syncthreads();
        // O += P V': k16-step ks covers keys kbase + 16 ks + 8 half + (0..7); A = P image group 2 ks + half, B = 16 bytes of each part of vt
#pragma unroll
        for (int ks = 0; ks < 2 * CW; ++ks) {
            const int aoff = ((2 * ks + half) * 32 + l31) * 16;
            const u32x4 ah = *reinterpret_cast<const u32x4*>(Pw + aoff);
            const u32x4 am = *reinterpret_cast<const u32x4*>(Pw + L::P_BYTES + aoff);
            const u32x4 al = *reinterpret_cast<const u32x4*>(Pw + 2 * L::P_BYTES + aoff);
            const size_t go = (size_t)(kbase / 8 + 2 * ks) * vgroup;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const u32x4 bh = *reinterpret_cast<const u32x4*>(vch[j] + go);
                const u32x4 bm = *reinterpret_cast<const u32x4*>(vch[j] + vpart + go);
                const u32x4 bl = *reinterpret_cast<const u32x4*>(vch[j] + 2 * vpart + go);
                acc[j] = td_mfma32_bf16(ah, bh, acc[j]);
                acc[j] = td_mfma32_bf16(ah, bm, acc[j]);
                acc[j] = td_mfma32_bf16(am, bh, acc[j]);
                acc[j] = td_mfma32_bf16(ah, bl, acc[j]);
                acc[j] = td_mfma32_bf16(am, bm, acc[j]);
                acc[j] = td_mfma32_bf16(al, bh, acc[j]);
            }
        }
    }
    // ---- row sums -> 1/l, epilogue (td_attn_h.h) -------------------------------------------------------------------
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
    if (ln && q0 < p.Lq) {
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

// ---- the 512-channel form: 64 queries and EIGHT waves per workgroup --------------------------------------------------------------------------------
// k_attention_b3<1, 4, 4> gives a v' fragment (16 bytes per lane, straight from global memory) to six MFMAs of ONE 32-query tile: a workgroup streams all of
// v' (6 bytes x Lk x 512) for 32 queries, ~53 B/clk/CU of vector-memory requests at the measured 474 us -- the CU's path delivers 64.  Here a workgroup holds
// TWO query tiles: wave w computes the score tile of query tile w >> 2 against key tile w & 3 of a 128-key super-tile (as before: one 32 x 32 tile per wave),
// and in P V' owns 64 channels (NT = 2) of BOTH query tiles, so every v' fragment feeds twelve MFMAs: half the vector-memory requests per MFMA, twice the
// LDS reads of P (which has the room).  One workgroup of eight waves per CU (96 KB of P images), two waves per SIMD as before.  The query parts live in LDS
// (24 KB; 48 registers otherwise), which pays for an explicit one-step-ahead request of the v' fragments -- the first under the softmax.
// Measured (profiles/r06ab_*): 32768 x 2048 x 512: 419 us against 465 on one box (440 before the explicit prefetch, against 474, on another); frame at 1024x2048 +1.8 %.
struct AttnLdsB3W {
    static constexpr int P_BYTES = 2 * 16 * 32 * 16;             // one part of one super-tile of P: [query tile 2][key group 16][32 queries][8 bf16]
    static constexpr int RED_FLOATS = 2 * 4 * 32;
    static constexpr int Q_BYTES = 2 * 4 * 3 * 1024;             // the pre-scaled query parts: [query tile 2][k16-step 4][part 3][lane 64][8 bf16]
    static constexpr int BYTES = 2 * 3 * P_BYTES + Q_BYTES + (2 * RED_FLOATS + 8 * 2 * 32) * 4;
};

TD_KERNEL void TD_LAUNCH_BOUNDS(512, 1) k_attention_b3w(AttnArgsB3 p) {
    constexpr int SK = 128, NT = 2;
    using L = AttnLdsB3W;
    TD_DYN_LDS(smem);
    char* Ps = smem;                                             // [2 buffers][3 parts][P_BYTES]
    char* Qs = Ps + 2 * 3 * L::P_BYTES;                          // the query parts (48 registers if kept there: what the v' prefetch below needs)
    float* red = reinterpret_cast<float*>(Qs + L::Q_BYTES);      // [query tile][key tile][32] tile-max exchange
    float* red2 = red + L::RED_FLOATS;                           // row-sum exchange
    float* scr = red2 + L::RED_FLOATS;                           // [8 waves][2 query tiles][32] rescale factors

    const int tid = threadIdx.x, lane = tid & 63, wave = TD_UNIFORM(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int qt = wave >> 2, kt = wave & 3;
    const int q0 = blockIdx.x * 64;
    const unsigned LDV = (unsigned)p.ldv;

    char* Qw = Qs + qt * (4 * 3 * 1024) + lane * 16;             // this lane's fragments of its score tile's query row: + (ks * 3 + part) * 1024
    {                                                            // wave kt splits k16-step kt of query tile qt (pre-scaled), once
        const int q = (q0 + qt * 32 + l31 < p.Lq) ? q0 + qt * 32 + l31 : p.Lq - 1;
        const float* src = p.q + (size_t)q * 64 + 16 * kt + 8 * half;
        u32x4 h, m, l;
        td_split3(td_ld4(src) * p.scale_log2e, td_ld4(src + 4) * p.scale_log2e, h, m, l);
        *reinterpret_cast<u32x4*>(Qw + (kt * 3 + 0) * 1024) = h;
        *reinterpret_cast<u32x4*>(Qw + (kt * 3 + 1) * 1024) = m;
        *reinterpret_cast<u32x4*>(Qw + (kt * 3 + 2) * 1024) = l;
    }
    __syncthreads();
    const int key_last = p.Lk - 1;
    auto load_k = [&](int kb, f32x4 (&kf)[8]) {
        const int key = (kb + l31 < p.Lk) ? kb + l31 : key_last;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* src = p.k + (size_t)key * 64 + 16 * ks + 8 * half;
            kf[2 * ks] = td_ld4(src);
            kf[2 * ks + 1] = td_ld4(src + 4);
        }
    };
    auto score_tile = [&](const f32x4 (&kf)[8]) -> f32x16 {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 kh, km, kl;
            const u32x4 qh = *reinterpret_cast<const u32x4*>(Qw + (ks * 3 + 0) * 1024), qm = *reinterpret_cast<const u32x4*>(Qw + (ks * 3 + 1) * 1024),
                        ql = *reinterpret_cast<const u32x4*>(Qw + (ks * 3 + 2) * 1024);
            td_split3(kf[2 * ks], kf[2 * ks + 1], kh, km, kl);
            s = td_mfma32_bf16(kh, qh, s);
            s = td_mfma32_bf16(kh, qm, s);
            s = td_mfma32_bf16(km, qh, s);
            s = td_mfma32_bf16(kh, ql, s);
            s = td_mfma32_bf16(km, qm, s);
            s = td_mfma32_bf16(kl, qh, s);
        }
        return s;
    };

    const int nsuper = (p.Lk + SK - 1) / SK;
    const float NEG = -3.0e38f;
    f32x16 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    float lsum = 0.f, rowmax[2] = {NEG, NEG};                     // references of BOTH query tiles (the accumulators of both live here); lsum: the score tile's
    const int cb0 = wave * (NT * 32);                             // first channel of this wave; tile j, lane l31 -> channel cb0 + 32 j + l31
    const unsigned short* vch[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) vch[j] = p.vt + ((size_t)half * LDV + cb0 + 32 * j + l31) * 8;
    const size_t vgroup = (size_t)LDV * 8;
    const size_t vpart = p.part_stride;

    f32x4 kf[8];
    load_k(kt * 32, kf);
    for (int st = 0; st < nsuper; ++st) {
        const int kbase = st * SK, kb = kbase + kt * 32;
        const f32x16 s = score_tile(kf);
        float lm = NEG;
        if (kb + 32 <= p.Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lm = __builtin_fmaxf(lm, s[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
                lm = (key < p.Lk && s[r] > lm) ? s[r] : lm;
            }
        }
        lm = fmaxf(lm, td_shfl_xor(lm, 32));
        if (half == 0) red[(qt * 4 + kt) * 32 + l31] = lm;
        __syncthreads();
        if (st + 1 < nsuper) load_k(kb + SK, kf);                 // next key tile, in flight under the P V' MFMAs
        u32x4 bq[2][NT][3];                                       // v' fragments of k16-step ks in bq[ks & 1]: requested one step ahead, the first here, under the softmax
                                                                  // (two steps ahead, three sets: 447 vs 419 us; left to the compiler -- six loads, then a wait for the first: 440)
        auto load_v = [&](int ks, u32x4 (&b)[NT][3]) {
            const size_t go = (size_t)(kbase / 8 + 2 * ks) * vgroup;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int part = 0; part < 3; ++part) b[j][part] = *reinterpret_cast<const u32x4*>(vch[j] + part * vpart + go);
        };
        load_v(0, bq[0]);
        TD_SCHED_FENCE();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float tm = NEG;
#pragma unroll
            for (int c = 0; c < 4; ++c) tm = fmaxf(tm, red[(t * 4 + c) * 32 + l31]);
            if (td_any(tm > rowmax[t] + 8.0f)) {                  // identical decision in all eight waves (same maxima, same references)
                const float nm = fmaxf(rowmax[t], tm);
                const float alpha = td_exp2(rowmax[t] - nm);
                if (t == qt) lsum *= alpha;
                rowmax[t] = nm;
                float* sc = scr + (wave * 2 + t) * 32;
                if (half == 0) sc[l31] = alpha;
                td_wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 a4 = td_ld4(sc + 8 * u + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[t][j][4 * u + e] *= a4[e];
                }
                td_wave_sync();
            }
        }
        const float ref = qt ? rowmax[1] : rowmax[0];
        char* Pb = Ps + (st & 1) * (3 * L::P_BYTES);
        char* Pw = Pb + qt * (16 * 512);
        f32x16 pr;
        if (kb + 32 <= p.Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = td_exp2(s[r] - ref);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[r] = (kb + 8 * (r >> 2) + 4 * half + (r & 3) < p.Lk) ? td_exp2(s[r] - ref) : 0.f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 h2, m2, l2;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                unsigned h_, m_, l_;
                td_split3_pair(pr[4 * g + 2 * e2], pr[4 * g + 2 * e2 + 1], h_, m_, l_);
                h2[e2] = h_; m2[e2] = m_; l2[e2] = l_;
                lsum += pr[4 * g + 2 * e2] + pr[4 * g + 2 * e2 + 1];
            }
            const int off = ((kt * 4 + g) * 32 + l31) * 16 + 8 * half;
            *reinterpret_cast<u32x2*>(Pw + off) = h2;
            *reinterpret_cast<u32x2*>(Pw + L::P_BYTES + off) = m2;
            *reinterpret_cast<u32x2*>(Pw + 2 * L::P_BYTES + off) = l2;
        }
        __syncthreads();
        // O += P V': k16-step ks covers keys kbase + 16 ks + 8 half + (0..7); A = the P images of both query tiles, group 2 ks + half; B = 16 bytes of each part of vt
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            TD_SCHED_FENCE();
            if (ks + 1 < 8) load_v(ks + 1, bq[(ks + 1) & 1]);
            u32x4 ah[2], am[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int aoff = t * (16 * 512) + ((2 * ks + half) * 32 + l31) * 16;
                ah[t] = *reinterpret_cast<const u32x4*>(Pb + aoff);
                am[t] = *reinterpret_cast<const u32x4*>(Pb + L::P_BYTES + aoff);
                al[t] = *reinterpret_cast<const u32x4*>(Pb + 2 * L::P_BYTES + aoff);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const u32x4 bh = bq[ks & 1][j][0], bm = bq[ks & 1][j][1], bl = bq[ks & 1][j][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t][j] = td_mfma32_bf16(ah[t], bh, acc[t][j]);
                    acc[t][j] = td_mfma32_bf16(ah[t], bm, acc[t][j]);
                    acc[t][j] = td_mfma32_bf16(am[t], bh, acc[t][j]);
                    acc[t][j] = td_mfma32_bf16(ah[t], bl, acc[t][j]);
                    acc[t][j] = td_mfma32_bf16(am[t], bm, acc[t][j]);
                    acc[t][j] = td_mfma32_bf16(al[t], bh, acc[t][j]);
                }
            }
        }
    }
    // ---- row sums -> 1/l, epilogue per query tile (k_attention_b3's) ----------------------------------------------
    lsum += td_shfl_xor(lsum, 32);
    if (half == 0) red2[(qt * 4 + kt) * 32 + l31] = lsum;
    __syncthreads();
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = p.bias ? p.bias[cb0 + 32 * j + l31] : 0.f;
    const bool ln = p.ln_part != nullptr;
    const TdBuf out_buf = td_make_buf(p.out, 0x80000000u);
    const TdBuf res_buf = td_make_buf(p.resid, p.resid ? 0x80000000u : 0u);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qb = q0 + 32 * t;
        float kshift[NT], s1[NT], s2[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) kshift[j] = s1[j] = s2[j] = 0.f;
        if (ln && qb < p.Lq) {
            float l = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) l += red2[(t * 4 + c) * 32];
            const float inv = 1.0f / l;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v = acc[t][j][0] * inv + bv[j];
                if (p.resid) v += p.resid[(size_t)qb * LDV + cb0 + 32 * j + l31];
                const float other = td_shfl_xor(v, 32);
                kshift[j] = half ? other : v;
            }
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float rv[4][NT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rg * 4 + i;
                const int q = qb + (r & 3) + 8 * (r >> 2) + 4 * half;
                const unsigned off = q < p.Lq ? ((unsigned)q * (unsigned)LDV + (unsigned)(cb0 + l31)) * 4u : TD_BUF_OOB;
#pragma unroll
                for (int j = 0; j < NT; ++j) rv[i][j] = td_buf_ld1(res_buf, off, (unsigned)(128 * j));
            }
            TD_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rg * 4 + i;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int q = qb + row;
                const bool live = q < p.Lq;
                const unsigned off = live ? ((unsigned)q * (unsigned)LDV + (unsigned)(cb0 + l31)) * 4u : TD_BUF_OOB;
                float l = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) l += red2[(t * 4 + c) * 32 + row];
                const float inv = 1.0f / l;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float o = acc[t][j][r] * inv + bv[j] + rv[i][j];
                    td_buf_st1(out_buf, off, (unsigned)(128 * j), o);
                    if (ln && live) { const float d = o - kshift[j]; s1[j] += d; s2[j] += d * d; }
                }
            }
        }
        const int strip = blockIdx.x * 2 + t;
        if (ln && strip < p.ln_nstr) {
            const int cnt = p.Lq - qb < 32 ? (p.Lq - qb > 0 ? p.Lq - qb : 0) : 32;
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
}

// V' [Lk][DV] fp32 -> the re-tiled, pre-split operand of k_attention_b3: 3 * DV * attn_lkpad(Lk) bf16
static inline void attn_prepare_vt_b3(const float* vp, int Lk, int DV, unsigned short* vt, hipStream_t s) {
    const int LkPad = attn_lkpad(Lk);
    TD_LAUNCH(k_attn_vt_b3, dim3(LkPad / 64, DV / 64), dim3(256), 64 * 65 * 4, s, vp, vt, Lk, LkPad, DV);
}
// form (DV = 512 k): 0 = by size -- the 64-query form needs a workgroup per CU to pay (Lq >= 16384; at Lq = 8192 it leaves half the chip idle: 179 vs 163 us) --,
// 1 = the 32-query form, 2 = the 64-query form (tests)
static inline int attn_launch_b3(const AttnArgs& a, int DV, unsigned short* vt, hipStream_t s, bool vt_ready = false, int form = 0) {
    const int LkPad = attn_lkpad(a.Lk);
    if (DV != 128 && (DV < 512 || DV % 512)) return -1;
    if (!vt_ready) attn_prepare_vt_b3(a.vp, a.Lk, DV, vt, s);
    AttnArgsB3 h;
    h.q = a.q; h.k = a.k; h.vt = vt; h.bias = a.bias; h.resid = a.resid; h.out = a.out; h.Lq = a.Lq; h.Lk = a.Lk; h.LkPad = LkPad;
    h.scale_log2e = a.scale_log2e; h.ln_part = a.ln_part; h.ldv = DV;
    h.part_stride = (size_t)(LkPad / 8) * DV * 8;
    if (DV == 128) {
        const int grid = (a.Lq + 63) / 64;
        h.ln_nstr = 2 * grid;
        TD_LAUNCH((k_attention_b3<2, 2, 2>), dim3(grid), dim3(256), (AttnLdsB3<2, 2>::BYTES), s, h);
        return 0;
    }
    const int grid = (a.Lq + 31) / 32;
    h.ln_nstr = grid;
    for (int c0 = 0; c0 < DV; c0 += 512) {
        AttnArgsB3 b = h;
        b.vt += (size_t)c0 * 8; b.out += c0;
        if (b.bias) b.bias += c0;
        if (b.resid) b.resid += c0;
        if (b.ln_part) b.ln_part += c0;
        if (form == 2 || (form == 0 && (a.Lq + 63) / 64 >= 256)) TD_LAUNCH(k_attention_b3w, dim3((a.Lq + 63) / 64), dim3(512), AttnLdsB3W::BYTES, s, b);
        else TD_LAUNCH((k_attention_b3<1, 4, 4>), dim3(grid), dim3(256), (AttnLdsB3<1, 4>::BYTES), s, b);
    }
    return 0;
}
