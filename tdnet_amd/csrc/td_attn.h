// td_attn.h -- fused attention-propagation kernel:  out = softmax(q k^T / sqrt(64)) v' + bias (+ resid)
//
// Reference: ScaledDotProductAttention (transformer.py:126-139: bmm, /8, softmax over keys, bmm) followed by the
// per-position fc of Attention.forward (transformer.py:84-86).  The fc is linear, so (P v) W^T + b = P (v W^T) + b: the
// host multiplies the (small, Lk-row) value matrix by W^T first (a 1x1 conv on the GEMM kernel) and this kernel consumes
// v' = v W^T; for the final propagation step (Lq = 16 Lk) that removes 16/17 of the fc FLOPs.  The residual argument
// fuses the "+ V_queue[i]" / "+ v_cur" adds of td4_psp18.py:146-151.
//
// The Lq x Lk score matrix is never materialised (the reference writes 268 MB of it per frame at 1024x2048).
// Two softmax schedules over the same tile code (template ONLINE, tdnet_opts.attention):
//   ONLINE = false: exact two-pass -- pass 1 computes every query's row maximum (QK^T only, 1/9 of the MFMA work), pass 2 recomputes
//                   the scores, exponentiates against the final maximum and accumulates P V'; no rescaling.
//   ONLINE = true : single pass.  softmax is invariant under the reference r subtracted in the exponent, so r need not be the row
//                   maximum -- it only has to keep exp2(s - r) inside fp32's range.  The kernel keeps a per-query reference and
//                   moves it (rescaling the accumulators and the running sum by exp2(r_old - r_new)) only when a tile's maximum
//                   exceeds it by more than TAU = 8, so P <= 2^8 and the rescale -- a wave-uniform branch -- runs a few times per
//                   query tile instead of once per key tile.  Results differ from the two-pass schedule by rounding only.
//   ONLINE = 2    : the same arithmetic, software-pipelined to ONE workgroup barrier per key super-tile (ONLINE = 1 needs two: tile
//                   maxima, then P): the scores of tile t+1 are computed, and their maxima published, before the barrier that
//                   publishes P of tile t, so the maxima are already visible when the next iteration needs them.
//
// MFMA mapping (fp32, v_mfma_f32_32x32x2_f32), one wave = 32 queries x (32*NT) output channels:
//   scores are computed TRANSPOSED, S^T = K Q^T (A = K tile, B = Q^T): lane (q = lane&31, half) then holds, in register
//   r, the score of ITS query against key (r&3) + 8 (r>>2) + 4 half -- four consecutive keys per 4 registers, which is
//   exactly one float4 of the P image the P V' MFMAs read back as their A operand.  Row max / row sum are per-lane
//   register reductions plus ONE cross-half shuffle; no serial-lane softmax.
//   A block is QW query tiles x CW channel groups of waves (QW*CW = 4).  The CW waves of a query tile split each
//   super-tile of 32*CW keys between them for S^T, publish P through LDS (conflict-free float4 rows), and each
//   accumulates its own 32*NT channels over all the keys.  V' rows are read straight from L2 into the B operand:
//   lane j owns NT consecutive channels, so one global float4 feeds 4 MFMAs.
#pragma once
#include "td_device.h"
#include "td_conv.h"   // td_ld4 / td_st4

// rows of the value matrix V' a launch may touch: Lk rounded up to the largest super-tile (128 keys); see load_v in the kernel
TD_HOSTDEV int attn_vp_rows(int Lk) { return (Lk + 127) / 128 * 128; }

struct AttnArgs {
    const float* q;      // [Lq][64]
    const float* k;      // [Lk][64]
    const float* vp;     // [Lk][DV]; CONTRACT: attn_vp_rows(Lk) rows allocated, the rows past Lk - 1 finite (zeros)
    const float* bias;   // [DV] or nullptr
    const float* resid;  // [Lq][DV] or nullptr
    float* out;          // [Lq][DV]
    int Lq, Lk;
    float scale_log2e;   // log2(e) / sqrt(d_k)
    // Optional: plane-LayerNorm strip statistics of the OUTPUT map (td4_psp18.py:306-312 normalises out over its Lq rows, per
    // channel).  Strip s = 32 consecutive query rows = one query tile: ln_part[s][DV] = mean of the strip, ln_part[ln_nstr + s][DV]
    // = sum (x - mean)^2 over it -- the layout k_ln_finalize combines exactly (td_misc.h).  nullptr = off.
    float* ln_part;
    int ln_nstr;         // number of strips = gridDim.x * QW
    // Row stride (floats) of vp / resid / out / ln_part.  A launch covers DV = 128 or 512 channels; a wider value matrix (td4 on a
    // Bottleneck backbone: d_v = 2048) is served as chunks of 512 channels, each its own launch on pointers offset by the chunk's
    // first channel and ldv = the full width (the scores are recomputed per chunk: 64 of 576 MACs per key).
    int ldv;
};

template <int QW, int CW>
struct AttnLds {
    static constexpr int P_FLOATS = QW * (8 * CW) * 32 * 4;      // one super-tile of P: [qw][kq][q][4]
    static constexpr int RED_FLOATS = QW * CW * 32;
    static constexpr int BYTES = (2 * P_FLOATS + 3 * RED_FLOATS) * 4;   // P (double buffered), row-max / row-sum exchange, per-wave rescale factors
};

#ifdef TD_ATTN_TRACE   // tools/attn_trace.hip only: s_memtime stamps of workgroups 0..7, every wave, the first 12 key super-tiles (ONLINE = 2):
// [wg][wave][st][0..4] = loop top, P written, next scores + maxima done, after the barrier, after the P V' MFMAs
#define TD_ATTN_STAMP(slot) do { if (blockIdx.x < 8 && st < 12 && lane == 0) \
    TD_ATTN_TRACE[(((size_t)blockIdx.x * 4 + wave) * 12 + st) * 5 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TD_ATTN_STAMP(slot) ((void)0)
#endif
template <int QW, int CW, int NT, int ONLINE>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_attention(AttnArgs p) {
    static_assert(QW * CW == 4, "4 waves per block");
    static_assert(NT == 4 || NT == 2, "lane owns NT consecutive channels: one float4 or one float2 of V' / the output");
    constexpr int DV = CW * NT * 32;
    constexpr int SK = 32 * CW;                                  // keys per super-tile
    using L = AttnLds<QW, CW>;
    TD_DYN_LDS(smem);
    float* Ps = reinterpret_cast<float*>(smem);                  // [2][P_FLOATS]
    float* red = Ps + 2 * L::P_FLOATS;                           // [QW][CW][32] row-max exchange
    float* red2 = red + L::RED_FLOATS;                           // [QW][CW][32] row-sum exchange
    float* scr = red2 + L::RED_FLOATS;                           // [4 waves][32] rescale factors of a wave's queries (ONLINE)

    const int tid = threadIdx.x, lane = tid & 63, wave = TD_UNIFORM(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int qw = wave / CW, cw = wave % CW;
    const int q0 = (blockIdx.x * QW + qw) * 32;                  // first query of this wave's tile
    if (gridDim.y > 1) {                                         // channel slices of DV in one launch: slice blockIdx.y of a p.ldv-wide value matrix
        const int c0 = blockIdx.y * DV;
        p.vp += c0; p.out += c0;
        if (p.bias) p.bias += c0;
        if (p.resid) p.resid += c0;
        if (p.ln_part) p.ln_part += c0;
    }

    // ---- this lane's query row, pre-scaled so that exp(s/8 - max) = exp2(S - M) -------------------------------
    // Out-of-range rows/keys are CLAMPED to the last valid one instead of branched around: the loads stay
    // unconditional (so they can be issued far ahead of their MFMAs) and the results are masked where they matter.
    f32x4 qf[8];
    {
        const int q = (q0 + l31 < p.Lq) ? q0 + l31 : p.Lq - 1;
#pragma unroll
        for (int g = 0; g < 8; ++g) qf[g] = td_ld4(p.q + (size_t)q * 64 + 8 * g + 4 * half) * p.scale_log2e;
    }
    const int key_last = p.Lk - 1;
    // K / V' tiles that lie wholly inside the matrices (all but the last, ragged one) are read with buffer loads whose per-lane
    // offset is a kernel constant and whose tile offset rides in an SGPR: no 64-bit address arithmetic and no clamps in the VALU
    // stream that shares the issue port with the MFMAs.  (Measured: this kernel does not wait on memory at all -- pointing every
    // K / V' load at L1-resident rows changes nothing -- it is bound by what the SIMDs issue.)
    const TdBuf kbuf = td_make_buf(p.k, (unsigned)p.Lk * 64u * 4u);
    const unsigned k_voff = ((unsigned)l31 * 64u + 4u * (unsigned)half) * 4u;
    auto load_k = [&](int kb, f32x4 (&kf)[8]) {
        if (kb + 32 <= p.Lk) {                                        // wave-uniform
#pragma unroll
            for (int g = 0; g < 8; ++g) kf[g] = td_buf_ld4(kbuf, k_voff, (unsigned)(kb * 64 + 8 * g) * 4u);
            return;
        }
        const int key = (kb + l31 < p.Lk) ? kb + l31 : key_last;
#pragma unroll
        for (int g = 0; g < 8; ++g) kf[g] = td_ld4(p.k + (size_t)key * 64 + 8 * g + 4 * half);
    };
    // S^T tile of 32 keys: this lane's 16 scores (query l31, keys kb + (r&3)+8(r>>2)+4half)
    auto score_tile = [&](const f32x4 (&kf)[8]) -> f32x16 {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = td_mfma32(kf[g][e], qf[g][e], s);
        return s;
    };

    const int nsuper = (p.Lk + SK - 1) / SK;
    const float NEG = -3.0e38f;

    // ---- pass 1 (two-pass schedule only): row maxima (next key tile in flight while the current one is multiplied) ---------
    float rowmax = NEG;                                           // reference subtracted in the exponent (ONLINE: moves lazily)
    if (!ONLINE) {
    float mx = NEG;
    {
        f32x4 kf[8], kn[8];
        load_k(cw * 32, kf);
        for (int st = 0; st < nsuper; ++st) {
            const int kb = st * SK + cw * 32;
            if (st + 1 < nsuper) load_k(kb + SK, kn);
            const f32x16 s = score_tile(kf);
            if (kb + 32 <= p.Lk) {                                    // wave-uniform: no key of this tile is masked
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, s[r]);   // v_max(3)_f32: half the instructions of compare + select
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
                    mx = (key < p.Lk && s[r] > mx) ? s[r] : mx;
                }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) kf[g] = kn[g];
        }
    }
    mx = fmaxf(mx, td_shfl_xor(mx, 32));
    if (half == 0) red[(qw * CW + cw) * 32 + l31] = mx;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CW; ++c) rowmax = fmaxf(rowmax, red[(qw * CW + c) * 32 + l31]);
    }

    // ---- pass 2: P = exp2(S - max), O += P V' ----------------------------------------------------------------
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float lsum = 0.f;
    const int cb = cw * (NT * 32) + l31 * NT;                     // this lane's first output channel
    // V' rows of k-group G of the super-tile starting at kbase: 4 keys per lane-half, one float4 (4 channels) each.  CONTRACT: vp has
    // attn_vp_rows(Lk) rows allocated (Lk rounded up to 128), the rows past Lk - 1 finite (zeros): a ragged last super-tile then
    // reads them like any other (P is exactly 0 for masked keys) and there is no "is this tile inside V'" test -- it used to sit
    // inside every one of the 18 calls per super-tile, a compare and two branches each in the middle of the MFMA stream.
    const unsigned LDV = (unsigned)p.ldv;
    const TdBuf vbuf = td_make_buf(p.vp, ((unsigned)(attn_vp_rows(p.Lk) - 1) * LDV + (unsigned)DV) * 4u);
    const unsigned v_voff = (4u * (unsigned)half * LDV + (unsigned)cb) * 4u;
    auto load_v = [&](int kbase, int G, f32x4 (&b)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned soff = (unsigned)(kbase + 8 * G + e) * LDV * 4u;
            if (NT == 4) b[e] = td_buf_ld4(vbuf, v_voff, soff);
            else {
                const f32x2 v2 = td_buf_ld2(vbuf, v_voff, soff);
                b[e][0] = v2[0]; b[e][1] = v2[1]; b[e][2] = 0.f; b[e][3] = 0.f;
            }
        }
    };
    if (ONLINE == 2) {
        // ---- single pass, one barrier per super-tile -------------------------------------------------------------
        auto tile_max = [&](const f32x16& s, int kb) -> float {
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
            return fmaxf(lm, td_shfl_xor(lm, 32));
        };
        float* redb[2] = {red, red2};                             // tile maxima, double buffered (red2 is free until the epilogue)
        f32x4 kf[8];
        load_k(cw * 32, kf);
        f32x16 s = score_tile(kf);
        {
            const float lm = tile_max(s, cw * 32);
            if (half == 0) redb[0][(qw * CW + cw) * 32 + l31] = lm;
        }
        if (nsuper > 1) load_k(SK + cw * 32, kf);
        __syncthreads();
        for (int st = 0; st < nsuper; ++st) {
            const int kbase = st * SK, kb = kbase + cw * 32;
            f32x4 bb[3][4];
            TD_ATTN_STAMP(0);
            load_v(kbase, 0, bb[0]);
            load_v(kbase, 1, bb[1]);
            // A: maxima of tile st (published by the previous barrier) -> reference -> P of tile st
            float tm = NEG;
#pragma unroll
            for (int c = 0; c < CW; ++c) tm = fmaxf(tm, redb[st & 1][(qw * CW + c) * 32 + l31]);
            if (td_any(tm > rowmax + 8.0f)) {
                const float nm = fmaxf(rowmax, tm);
                const float alpha = td_exp2(rowmax - nm);
                lsum *= alpha;
                rowmax = nm;
                float* sc = scr + wave * 32;
                if (half == 0) sc[l31] = alpha;
                td_wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 a4 = td_ld4(sc + 8 * u + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[j][4 * u + e] *= a4[e];
                }
                td_wave_sync();
            }
            float* Pw = Ps + (st & 1) * L::P_FLOATS + qw * (8 * CW * 128);
            f32x16 pr;
            if (kb + 32 <= p.Lk) {                                    // wave-uniform: no key of this tile is masked
#pragma unroll
                for (int r = 0; r < 16; ++r) { pr[r] = td_exp2(s[r] - rowmax); lsum += pr[r]; }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
                    pr[r] = (key < p.Lk) ? td_exp2(s[r] - rowmax) : 0.f;
                    lsum += pr[r];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 v = {pr[4 * u], pr[4 * u + 1], pr[4 * u + 2], pr[4 * u + 3]};
                td_st4(Pw + ((cw * 8 + 2 * u + half) * 32 + l31) * 4, v);
            }
            TD_ATTN_STAMP(1);
            // B: scores of tile st+1 and their maxima, before the barrier
            if (st + 1 < nsuper) {
                s = score_tile(kf);
                const float lm = tile_max(s, kb + SK);
                if (half == 0) redb[(st + 1) & 1][(qw * CW + cw) * 32 + l31] = lm;
                if (st + 2 < nsuper) load_k(kb + 2 * SK, kf);     // in flight under the P V' MFMAs below
            }
            TD_ATTN_STAMP(2);
            __syncthreads();
            TD_ATTN_STAMP(3);
#pragma unroll
            for (int G = 0; G < 4 * CW; ++G) {
                if (G + 2 < 4 * CW) load_v(kbase, G + 2, bb[(G + 2) % 3]);
                const f32x4 a4 = td_ld4(Pw + ((2 * G + half) * 32 + l31) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = td_mfma32(a4[e], bb[G % 3][e][j], acc[j]);
                if (G + 2 < 4 * CW) { TD_SCHED_GROUP(0x020, 4); TD_SCHED_GROUP(0x100, 1); }
                TD_SCHED_GROUP(0x008, 4 * NT);
            }
            TD_ATTN_STAMP(4);
        }
    } else {
    f32x4 kf[8];
    load_k(cw * 32, kf);
    for (int st = 0; st < nsuper; ++st) {
        const int kbase = st * SK, kb = kbase + cw * 32;
        f32x4 bb[3][4];                                           // V' groups G+1, G+2 in flight while G is multiplied
        load_v(kbase, 0, bb[0]);                                  // in flight under the 32 score MFMAs
        load_v(kbase, 1, bb[1]);
        const f32x16 s = score_tile(kf);
        if (ONLINE) {
            // this wave's tile maximum per query -> the query tile's maximum over the CW waves -> move the reference if needed
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
            float tm = NEG;
#pragma unroll
            for (int c = 0; c < CW; ++c) tm = fmaxf(tm, red[(qw * CW + c) * 32 + l31]);
            // Every wave of the query tile reads the same maxima and holds the same references, so they all take this branch together.
            if (td_any(tm > rowmax + 8.0f)) {
                const float nm = fmaxf(rowmax, tm);
                const float alpha = td_exp2(rowmax - nm);             // first tile: exp2(-3e38 - nm) = 0 on zero accumulators
                lsum *= alpha;
                rowmax = nm;
                float* sc = scr + wave * 32;                          // wave-private: accumulator row i belongs to query i, lane i holds its factor
                if (half == 0) sc[l31] = alpha;
                td_wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 a4 = td_ld4(sc + 8 * u + 4 * half);   // rows 8u + 4 half + {0..3} = registers 4u + {0..3}
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[j][4 * u + e] *= a4[e];
                }
                td_wave_sync();
            }
        }
        f32x16 pr;
        if (kb + 32 <= p.Lk) {                                        // wave-uniform: no key of this tile is masked
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = td_exp2(s[r] - rowmax);
                pr[r] = e;
                lsum += e;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float e = (key < p.Lk) ? td_exp2(s[r] - rowmax) : 0.f;
                pr[r] = e;
                lsum += e;
            }
        }
        float* Pw = Ps + (st & 1) * L::P_FLOATS + qw * (8 * CW * 128);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 v = {pr[4 * u], pr[4 * u + 1], pr[4 * u + 2], pr[4 * u + 3]};
            td_st4(Pw + ((cw * 8 + 2 * u + half) * 32 + l31) * 4, v);
        }
        __syncthreads();
        if (st + 1 < nsuper) load_k(kb + SK, kf);                 // next key tile, hidden under the P V' MFMAs
#pragma unroll
        for (int G = 0; G < 4 * CW; ++G) {
            if (G + 2 < 4 * CW) load_v(kbase, G + 2, bb[(G + 2) % 3]);
            const f32x4 a4 = td_ld4(Pw + ((2 * G + half) * 32 + l31) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = td_mfma32(a4[e], bb[G % 3][e][j], acc[j]);
            if (G + 2 < 4 * CW) { TD_SCHED_GROUP(0x020, 4); TD_SCHED_GROUP(0x100, 1); }
            TD_SCHED_GROUP(0x008, 4 * NT);
        }
    }
    }
    // ---- row sums -> 1/l, epilogue ---------------------------------------------------------------------------
    lsum += td_shfl_xor(lsum, 32);
    if (half == 0) red2[(qw * CW + cw) * 32 + l31] = lsum;
    __syncthreads();
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = p.bias[cb + j];
    }
    // Optional LayerNorm strip statistics of the finished rows (strip = this query tile), in ONE sweep and without keeping the rows:
    // sums of (x - K) and (x - K)^2 against a per-channel shift K = the strip's first row (a sample of the data, so the one-pass
    // formula M2 = S2 - S1^2 / n loses about a bit instead of cancelling catastrophically when |mean| >> spread); fixed summation order.
    const bool ln = p.ln_part != nullptr;
    f32x4 kshift = {0.f, 0.f, 0.f, 0.f}, s1 = kshift, s2 = kshift;
    if (ln && q0 < p.Lq) {                                            // row 0 of the tile lives in register 0 of lane-half 0
        float l = 0.f;
#pragma unroll
        for (int c = 0; c < CW; ++c) l += red2[(qw * CW + c) * 32];
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < NT; ++j) kshift[j] = acc[j][0] * inv + bv[j];
        if (p.resid) {
#pragma unroll
            for (int j = 0; j < NT; ++j) kshift[j] += p.resid[(size_t)q0 * LDV + cb + j];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float other = td_shfl_xor(kshift[j], 32);
            kshift[j] = half ? other : kshift[j];
        }
    }
    // Stores and residual loads go through buffer descriptors (32-bit offsets, rows >= Lq get the out-of-range offset and are dropped /
    // read as zero); the residual rows are requested eight at a time BEFORE the first is used -- a load, a wait and a store per row
    // exposes the memory latency sixteen times.  No residual: a descriptor with zero records (loads return 0, no memory access).
    const TdBuf out_buf = td_make_buf(p.out, 0x80000000u);
    const TdBuf res_buf = td_make_buf(p.resid, p.resid ? 0x80000000u : 0u);
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
        f32x4 rv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = rg * 8 + i;
            const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const unsigned off = q < p.Lq ? ((unsigned)q * (unsigned)LDV + (unsigned)cb) * 4u : TD_BUF_OOB;
            if (NT == 4) rv[i] = td_buf_ld4(res_buf, off, 0u);
            else { const f32x2 t = td_buf_ld2(res_buf, off, 0u); rv[i] = f32x4{t[0], t[1], 0.f, 0.f}; }
        }
        TD_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = rg * 8 + i;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int q = q0 + row;
            const bool live = q < p.Lq;
            const unsigned off = live ? ((unsigned)q * (unsigned)LDV + (unsigned)cb) * 4u : TD_BUF_OOB;
            float l = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) l += red2[(qw * CW + c) * 32 + row];
            const float inv = 1.0f / l;
            if (NT == 4) {
                f32x4 o = {acc[0][r] * inv, acc[1][r] * inv, acc[NT - 2][r] * inv, acc[NT - 1][r] * inv};
                o = o + bv;
                o = o + rv[i];
                td_buf_st4(out_buf, off, o);
                if (ln && live) { const f32x4 d = o - kshift; s1 = s1 + d; s2 = s2 + d * d; }
            } else {
                f32x2 o = {acc[0][r] * inv + bv[0], acc[1][r] * inv + bv[1]};
                o[0] += rv[i][0]; o[1] += rv[i][1];
                td_buf_st2(out_buf, off, 0u, o);
                if (ln && live) {
                    const float d0 = o[0] - kshift[0], d1 = o[1] - kshift[1];
                    s1[0] += d0; s1[1] += d1; s2[0] += d0 * d0; s2[1] += d1 * d1;
                }
            }
        }
    }
    if (ln) {
        const int cnt = p.Lq - q0 < 32 ? (p.Lq - q0 > 0 ? p.Lq - q0 : 0) : 32;
        f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = mean;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float t1 = s1[j] + td_shfl_xor(s1[j], 32), t2 = s2[j] + td_shfl_xor(s2[j], 32);
            if (cnt) {
                const float rn = 1.0f / (float)cnt;
                mean[j] = kshift[j] + t1 * rn;
                const float v = t2 - t1 * t1 * rn;
                m2[j] = v > 0.f ? v : 0.f;
            }
        }
        if (half == 0) {
            const int strip = blockIdx.x * QW + qw;
            float* pm = p.ln_part + (size_t)strip * LDV + cb;
            float* pq = p.ln_part + ((size_t)p.ln_nstr + strip) * LDV + cb;
            if (NT == 4) { td_st4(pm, mean); td_st4(pq, m2); }
            else { pm[0] = mean[0]; pm[1] = mean[1]; pq[0] = m2[0]; pq[1] = m2[1]; }
        }
    }
}

// query tiles (= LayerNorm strips) of a launch
static inline int attn_strips(int Lq, int DV) { return DV % 512 == 0 ? (Lq + 31) / 32 : 2 * ((Lq + 63) / 64); }

// `slices` (DV = 512 only): the launch is split into two 256-channel slices (grid.y = 2, <1,4,2>: a lane owns 2 channels) -- for the
// cached-frame steps of the propagation chain (Lq = Lk = 2048 at 1024x2048: 64 query tiles are 64 workgroups on 256 CUs; with two
// slices 128, each with half the P V' work; the 64-wide Q K^T is computed by both).
static inline int attn_launch(AttnArgs a, int DV, int online, hipStream_t s, bool slices = false) {
    a.ldv = DV;
    if (DV == 512 && slices && !a.ln_part) {
        const dim3 grid((a.Lq + 31) / 32, 2);
        if (online == 2) TD_LAUNCH((k_attention<1, 4, 2, 2>), grid, dim3(256), (AttnLds<1, 4>::BYTES), s, a);
        else if (online) TD_LAUNCH((k_attention<1, 4, 2, 1>), grid, dim3(256), (AttnLds<1, 4>::BYTES), s, a);
        else TD_LAUNCH((k_attention<1, 4, 2, 0>), grid, dim3(256), (AttnLds<1, 4>::BYTES), s, a);
        return 0;
    }
    if (DV >= 512 && DV % 512 == 0) {
        const int grid = (a.Lq + 31) / 32;
        a.ln_nstr = grid;
        for (int c0 = 0; c0 < DV; c0 += 512) {                         // one launch per 512 channels (DV = 512: a single one)
            AttnArgs b = a;
            b.vp += c0; b.out += c0;
            if (b.bias) b.bias += c0;
            if (b.resid) b.resid += c0;
            if (b.ln_part) b.ln_part += c0;
            if (online == 2) TD_LAUNCH((k_attention<1, 4, 4, 2>), dim3(grid), dim3(256), (AttnLds<1, 4>::BYTES), s, b);
            else if (online) TD_LAUNCH((k_attention<1, 4, 4, 1>), dim3(grid), dim3(256), (AttnLds<1, 4>::BYTES), s, b);
            else TD_LAUNCH((k_attention<1, 4, 4, 0>), dim3(grid), dim3(256), (AttnLds<1, 4>::BYTES), s, b);
        }
    } else if (DV == 128) {
        // two query tiles x two channel halves: 2 waves per SIMD at Lq = 32768 (<4,1,4> -- four tiles, all channels -- runs one)
        const int grid = (a.Lq + 63) / 64;
        a.ln_nstr = 2 * grid;
        if (online == 2) TD_LAUNCH((k_attention<2, 2, 2, 2>), dim3(grid), dim3(256), (AttnLds<2, 2>::BYTES), s, a);
        else if (online) TD_LAUNCH((k_attention<2, 2, 2, 1>), dim3(grid), dim3(256), (AttnLds<2, 2>::BYTES), s, a);
        else TD_LAUNCH((k_attention<2, 2, 2, 0>), dim3(grid), dim3(256), (AttnLds<2, 2>::BYTES), s, a);
    } else {
        return -1;
    }
    return 0;
}
