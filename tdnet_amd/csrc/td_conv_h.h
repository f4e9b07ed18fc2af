// td_conv_h.h -- the same NHWC implicit-GEMM convolution on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate):
// the "fp16 MFMA" mode of BASELINE.json config 5.  Weights are packed as fp16 on the host.  Activations (template IN16 / OUT16):
//   IN16 = false: the input map is fp32 in HBM; the gather converts it to fp16 (round-to-nearest-even) on the way into LDS;
//   IN16 = true : the input map (and the residual) is fp16 NHWC in HBM -- half the gather bytes, one 16-byte load per LDS slot, no
//                 converts in the loop;
//   OUT16       : the epilogue rounds act(acc + bias + resid) to fp16 and stores 8 bytes per lane instead of 16.
// Inside the backbone every map between two convs is fp16 (IN16 = OUT16 = true); the convs at its rim read or write fp32 (the stem's
// output, c4 for the pyramid / Encoding / head, which keep fp32 storage).
//
// Per K step (BK = 64 channels of one tap) a 128x128 block issues 64 MFMAs of 32 cycles each instead of 256 of 64 cycles:
// 16x the matrix rate, so the kernel lives or dies by how little else it does per step -- one ds_read_b128 per operand per
// MFMA (LDS image [kq = 8 groups of 8 k][row][8 halfs], conflict free), 12 buffer loads + 16 packed converts + 8 LDS
// writes per thread, all spread between the MFMAs of the two-stage pipeline of td_conv.h.
//
// Numerics: inputs are rounded to fp16 (2^-11 relative), products are exact in fp32, accumulation is fp32.  This mode does
// NOT meet the 1e-3 logits gate of the fp32 path; bench.py reports its parity next to the number.
#pragma once
#include "td_conv.h"

template <int BM, int BN>
struct ConvLdsH {                                   // sizes in halfs
    static constexpr int A_STRIDE = BM * 8 + 8;     // per k-group: BM rows x 8 halfs (+16 B pad, conflict-free ds_write_b128)
    static constexpr int B_STRIDE = BN * 8;
    static constexpr int A_HALFS = 8 * A_STRIDE;
    static constexpr int B_HALFS = 8 * B_STRIDE;
    static constexpr int BUF_HALFS = A_HALFS + B_HALFS;
    static constexpr int BYTES = 2 * BUF_HALFS * 2;
};

TD_DEV f16x8 td_ld8h(const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); }
TD_DEV void td_st8h(_Float16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
TD_DEV f16x8 td_cvt8(f32x4 lo, f32x4 hi) {
    f16x8 r;
    r[0] = (_Float16)lo[0]; r[1] = (_Float16)lo[1]; r[2] = (_Float16)lo[2]; r[3] = (_Float16)lo[3];
    r[4] = (_Float16)hi[0]; r[5] = (_Float16)hi[1]; r[6] = (_Float16)hi[2]; r[7] = (_Float16)hi[3];
    return r;
}

// Epilogue of the fp16-MFMA kernels: out[m][n] = act(acc + bias[n] (+ resid[m][n])), resid fp16 when RES16, out fp16 when OUT16.
// Same lane-pair exchange as td_store_acc (NT == 2): a lane ends up with 4 consecutive channels of one row.
template <int MT, int NT, bool OUT16, bool RES16>
TD_DEV void td_store_acc_h(const f32x16 (&acc)[MT][NT], void* outv, const float* bias, const void* residv, int M, int N, int act,
                           int m_base, int n_base, int lane) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int half = lane >> 5, l31 = lane & 31;
    const float slope = td_act_slope(act);
    auto activate = [&](float v) { return td_activate(v, slope); };
    float* outf = reinterpret_cast<float*>(outv);
    _Float16* outh = reinterpret_cast<_Float16*>(outv);
    const float* resf = reinterpret_cast<const float*>(residv);
    const _Float16* resh = reinterpret_cast<const _Float16*>(residv);
    if (NT == 2 && (N & 3) == 0 && ((((size_t)outv) | ((size_t)residv)) & 15) == 0 && (size_t)(M + 128) * N < (1u << 29)) {   // wave-uniform
        // as td_store_acc16 (td_conv.h): buffer-addressed, the 8 residual vectors of a 32-row group requested together, no predicates
        const int odd = l31 & 1;
        const int chan = n_base + 4 * (l31 >> 1);
        const bool cok = chan < N;
        constexpr unsigned EO = OUT16 ? 2u : 4u, ER = RES16 ? 2u : 4u;
        const unsigned elems = (unsigned)M * (unsigned)N;
        const TdBuf out_buf = td_make_buf(outf, elems * EO);
        const TdBuf res_buf = td_make_buf(resf, residv ? elems * ER : 0u);
        const TdBuf bias_buf = td_make_buf(bias, (unsigned)N * 4u);
        f32x4 bv;
        if ((((size_t)bias) & 15) == 0) bv = td_buf_ld4(bias_buf, cok ? (unsigned)chan * 4u : TD_BUF_OOB, 0u);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = td_buf_ld1(bias_buf, cok ? (unsigned)(chan + e) * 4u : TD_BUF_OOB, 0u);
        }
        TD_PIN(bv);
        const unsigned e0 = (unsigned)(m_base + 4 * half + odd) * (unsigned)N + (unsigned)chan;
        const unsigned base_o = cok ? e0 * EO : TD_BUF_OOB, base_r = cok ? e0 * ER : TD_BUF_OOB;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x4 rv[8];
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const int r = 2 * rp;
                const unsigned rows = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)N;
                if (RES16) {
                    const f16x4 rh = __builtin_bit_cast(f16x4, td_buf_ld2(res_buf, base_r + rows * ER, 0u));
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[rp][e] = (float)rh[e];
                } else rv[rp] = td_buf_ld4(res_buf, base_r + rows * ER, 0u);
            }
            TD_SCHED_FENCE();
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const int r = 2 * rp;
                const float a0 = acc[i][0][r], a1 = acc[i][NT - 1][r], c0 = acc[i][0][r + 1], c1 = acc[i][NT - 1][r + 1];
                const float x = td_swap1(odd ? a0 : c0), y = td_swap1(odd ? a1 : c1);
                f32x4 v;
                if (odd) { v[0] = x; v[1] = y; v[2] = c0; v[3] = c1; }
                else     { v[0] = a0; v[1] = a1; v[2] = x; v[3] = y; }
                v = v + bv + rv[rp];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = activate(v[e]);
                const unsigned rows = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)N;
                if (OUT16) {
                    f16x4 oh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) oh[e] = (_Float16)v[e];
                    td_buf_st2(out_buf, base_o + rows * EO, 0u, __builtin_bit_cast(f32x2, oh));
                } else td_buf_st4(out_buf, base_o + rows * EO, v);
                TD_SCHED_FENCE();
            }
        }
        return;
    }
    const int nb = n_base + l31 * NT;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= M) continue;
            const size_t o = (size_t)m * N + nb;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (nb + j >= N) continue;
                float v = acc[i][j][r] + bias[nb + j];
                if (residv) v += RES16 ? (float)resh[o + j] : resf[o + j];
                v = activate(v);
                if (OUT16) outh[o + j] = (_Float16)v; else outf[o + j] = v;
            }
        }
    }
}

// ConvArgs as td_conv.h, except: wp = packed fp16 weights [nsteps][8][CoutPad][8 halfs], nsteps = (Cin/64)*KS*KS; with IN16 `in` and
// `resid` point at fp16 maps, with OUT16 `out` does.
//
// STEM (the 7x7 stride-2 conv on the 4-channel padded image, fp32 in HBM): a 16-byte LDS slot = 8 halfs = the 4 channels of TWO
// horizontally adjacent taps, a K step = 8 slots x 2 = 16 taps = two kernel rows of 8 (7 + one zero-weight column), 4 steps = rows
// 0..7 (row 7: zero weights).  K = 256 for 147 products -- 57 % useful, on a pipe 16x faster than the fp32 one.
// The body takes its block index as an argument (bid of nblk): k_conv_igemm_h runs one conv per launch, k_conv_igemm_h_group up to three.
template <int BM, int BN, int WGM, int WGN, int KS, bool IN16, bool OUT16, bool STEM = false>
TD_DEV void conv_igemm_h_body(const ConvArgs& p, int bid, int nblk) {
    static_assert(!STEM || (!IN16 && KS == 7), "the stem reads the fp32 image");
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int AL = BM / 32, BL = BN / 32;          // 16-byte LDS slots per thread per step (A: 8 channels of one pixel)
    static_assert((AL == 2 || AL == 4) && (BL == 2 || BL == 4), "staging slots are spread over the 4 k-groups");
    using L = ConvLdsH<BM, BN>;
    constexpr int NTAPS = KS * KS;
    constexpr int AR = IN16 ? AL : 2 * AL;              // staging registers (16 bytes each) per A tile
    constexpr unsigned EB = IN16 ? 2u : 4u;             // bytes per input element
    TD_DYN_LDS(smem);
    _Float16* lds = reinterpret_cast<_Float16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lin = td_xcd_remap(bid, nblk);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int a_row = tid >> 3, a_kq = tid & 7;         // slot i -> row a_row + 32 i, channels a_kq*8 .. +7 of the 64-channel chunk
    int a_by[AL], a_bx[AL];
    unsigned a_off[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + a_row + 32 * i;
        const int oy = m / p.Wo, ox = m - oy * p.Wo;
        a_by[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 28);
        a_bx[i] = ox * p.stride - p.pad;
        if constexpr (STEM) {                           // slot a_kq = taps (ky, kx), (ky, kx + 1) with ky = a_kq / 4 (+ 2 per step), kx = 2 (a_kq % 4)
            a_by[i] += a_kq >> 2;
            a_bx[i] += 2 * (a_kq & 3);
            a_off[i] = ((unsigned)a_by[i] * (unsigned)p.W + (unsigned)a_bx[i]) * 16u;
        } else
        a_off[i] = (((unsigned)a_by[i] * (unsigned)p.W + (unsigned)a_bx[i]) * (unsigned)p.Cin + (unsigned)a_kq * 8u) * EB;
    }
    unsigned a_taps[AL];                                // bit t: tap t of slot i reads inside the image (td_conv.h)
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        a_taps[i] = 0u;
        if (!STEM) {
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                const int iy = a_by[i] + (t / KS) * p.dil, ix = a_bx[i] + (t % KS) * p.dil;
                a_taps[i] |= ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (1u << t) : 0u;
            }
        }
    }
    const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * EB);
    const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
    unsigned b_off[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        b_off[i] = (unsigned)(kq * p.CoutPad + n0 + n) * 16u;
    }
    const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;

    int l_step = 0, l_chunk = 0, l_tap = 0;
    auto load_tile = [&](f32x4 (&ra)[AR], f32x4 (&rb)[BL]) {
        if constexpr (STEM) {
            const unsigned delta = (unsigned)(2 * l_step * p.W) * 16u;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int iy = a_by[i] + 2 * l_step, ix = a_bx[i];
                const bool oky = (unsigned)iy < (unsigned)p.H;
                ra[2 * i] = td_buf_ld4(in_buf, oky && (unsigned)ix < (unsigned)p.W ? a_off[i] + delta : TD_BUF_OOB, 0u);
                ra[2 * i + 1] = td_buf_ld4(in_buf, oky && (unsigned)(ix + 1) < (unsigned)p.W ? a_off[i] + delta + 16u : TD_BUF_OOB, 0u);
            }
            const unsigned wsoff = (unsigned)(l_step < p.nsteps ? l_step : p.nsteps - 1) * w_step_bytes;
#pragma unroll
            for (int i = 0; i < BL; ++i) rb[i] = td_buf_ld4(w_buf, b_off[i], wsoff);
            ++l_step;
            return;
        }
        const int ky = l_tap / KS;
        const int dy = ky * p.dil, dx = (l_tap - ky * KS) * p.dil;
        const unsigned delta = (unsigned)((dy * p.W + dx) * p.Cin + l_chunk * 64) * EB;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const bool ok = ((a_taps[i] >> l_tap) & 1u) != 0u;
            const unsigned off = ok ? a_off[i] + delta : TD_BUF_OOB;
            if constexpr (IN16) ra[i] = td_buf_ld4(in_buf, off, 0u);  // 8 fp16 channels = the LDS slot as it is
            else {
                ra[2 * i] = td_buf_ld4(in_buf, off, 0u);
                ra[2 * i + 1] = td_buf_ld4(in_buf, ok ? off + 16u : TD_BUF_OOB, 0u);
            }
        }
        const unsigned wsoff = (unsigned)(l_step < p.nsteps ? l_step : p.nsteps - 1) * w_step_bytes;
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = td_buf_ld4(w_buf, b_off[i], wsoff);
        ++l_step;
        if (++l_tap == NTAPS) { l_tap = 0; ++l_chunk; }
    };
    auto store_a = [&](int buf, int i, const f32x4 (&ra)[AR]) {
        _Float16* dst = lds + buf * L::BUF_HALFS + a_kq * L::A_STRIDE + (a_row + 32 * i) * 8;
        if constexpr (IN16) *reinterpret_cast<f32x4*>(dst) = ra[i];
        else td_st8h(dst, td_cvt8(ra[2 * i], ra[2 * i + 1]));
    };
    auto store_b = [&](int buf, int i, const f32x4 (&rb)[BL]) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        *reinterpret_cast<f32x4*>(lds + buf * L::BUF_HALFS + L::A_HALFS + kq * L::B_STRIDE + n * 8) = rb[i];   // already fp16 bits
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one K step = 4 MFMA k-steps of 16; k-step g uses LDS groups 2g (lanes 0-31) and 2g+1 (lanes 32-63)
    auto compute = [&](int buf, const f32x4 (&sa)[AR], const f32x4 (&sb)[BL]) {
        const _Float16* As = lds + buf * L::BUF_HALFS + (wm * WM + l31) * 8;
        const _Float16* Bs = lds + buf * L::BUF_HALFS + L::A_HALFS + (wn * WN + l31) * 8;
        f16x8 af[2][MT], bf[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = td_ld8h(As + half * L::A_STRIDE + i * 256);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[0][j] = td_ld8h(Bs + half * L::B_STRIDE + j * 256);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[(g + 1) & 1][i] = td_ld8h(As + (2 * g + 2 + half) * L::A_STRIDE + i * 256);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[(g + 1) & 1][j] = td_ld8h(Bs + (2 * g + 2 + half) * L::B_STRIDE + j * 256);
            }
#pragma unroll
            for (int i = 0; i < AL; ++i) if (i * (4 / AL) == g) store_a(buf ^ 1, i, sa);
#pragma unroll
            for (int i = 0; i < BL; ++i) if (i * (4 / BL) == g) store_b(buf ^ 1, i, sb);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = td_mfma32_f16(af[g & 1][i], bf[g & 1][j], acc[i][j]);
        }
    };

    // two-stage pipeline, branch free (see td_conv.h): tile s+2 in flight, tile s+1 written to LDS under the MFMAs of tile s
    f32x4 ra[AR], rb[BL], ra2[AR], rb2[BL];
    load_tile(ra, rb);
#pragma unroll
    for (int i = 0; i < AL; ++i) store_a(0, i, ra);
#pragma unroll
    for (int i = 0; i < BL; ++i) store_b(0, i, rb);
    load_tile(ra, rb);
    __syncthreads();
    int step = 0;                                       // whole periods, then the odd last step (no break inside the loop: td_conv.h)
    for (; step + 1 < p.nsteps; step += 2) {
        load_tile(ra2, rb2);
        compute(0, ra, rb);
        __syncthreads();
        load_tile(ra, rb);
        compute(1, ra2, rb2);
        __syncthreads();
    }
    if (step < p.nsteps) compute(0, ra, rb);            // stages a surplus tile into buffer 1, never read

    td_store_acc_h<MT, NT, OUT16, IN16>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * WM, n0 + wn * WN, lane);
}
template <int BM, int BN, int WGM, int WGN, int KS, bool IN16, bool OUT16, bool STEM = false>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_conv_igemm_h(ConvArgs p) {
    conv_igemm_h_body<BM, BN, WGM, WGN, KS, IN16, OUT16, STEM>(p, blockIdx.x, gridDim.x);
}
// Up to three INDEPENDENT convs of one kernel form in ONE launch (round 5): blocks [0, end[0]) run c[0], [end[0], end[1]) c[1], the rest c[2].
// The Encoding's five 1x1 convs are two such groups (value / query / key first layers on z; query / key second layers): on a 10^4-pixel
// map each of them is a launch of 6 .. 340 workgroups that lasts 4 - 9 us, most of it latency; side by side they fill the chip once.  The
// arguments of the block's conv are selected with scalar moves (blockIdx is uniform), the body is compiled once.  Same products in the same
// order as k_conv_igemm_h: bit-identical outputs.
struct ConvGroupArgs {
    ConvArgs c[3];
    int end[3];
};
template <int BM, int BN, int WGM, int WGN, int KS0, int KS1, bool IN16, bool OUT16>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_conv_igemm_h_group(ConvGroupArgs q) {
    const int b = blockIdx.x;
    const int g = b < q.end[0] ? 0 : b < q.end[1] ? 1 : 2;
    const int first = g == 0 ? 0 : g == 1 ? q.end[0] : q.end[1];
    const int last = g == 0 ? q.end[0] : g == 1 ? q.end[1] : q.end[2];
    if constexpr (KS0 == KS1) {
        const ConvArgs p = g == 0 ? q.c[0] : g == 1 ? q.c[1] : q.c[2];
        conv_igemm_h_body<BM, BN, WGM, WGN, KS0, IN16, OUT16, false>(p, b - first, last - first);
    } else if (g == 0) {                                             // c[0] with another kernel size (a block's 3x3 conv1 beside its 1x1 downsample)
        conv_igemm_h_body<BM, BN, WGM, WGN, KS0, IN16, OUT16, false>(q.c[0], b, q.end[0]);
    } else {
        const ConvArgs p = g == 1 ? q.c[1] : q.c[2];
        conv_igemm_h_body<BM, BN, WGM, WGN, KS1, IN16, OUT16, false>(p, b - first, last - first);
    }
}

static inline int conv_nsteps_h(int Cin, int KS) { return (Cin / 64) * KS * KS; }

// fp16 packing: [step][kq][slot][8 halfs]; slot -> output channel exactly as conv_pack_weights, k = chunk*64 + kq*8 + e.
static inline void conv_pack_weights_h(const float* w, int Cout, int Cin, int KS, ConvTile tile, _Float16* dst) {
    const ConvTileDims d = conv_tile_dims(tile);
    const int WN = d.BN / d.WGN, NT = WN / 32;
    const int CoutPad = conv_cout_pad(Cout, tile);
    const int nsteps = conv_nsteps_h(Cin, KS), ntaps = KS * KS;
    for (int step = 0; step < nsteps; ++step)
        for (int kq = 0; kq < 8; ++kq)
            for (int slot = 0; slot < CoutPad; ++slot) {
                const int tn = slot / d.BN, within = slot % d.BN, wn = within / WN, w2 = within % WN;
                const int nt = w2 / 32, j = w2 % 32;
                const int n = tn * d.BN + wn * WN + j * NT + nt;
                _Float16* o = dst + (((size_t)step * 8 + kq) * CoutPad + slot) * 8;
                const int chunk = step / ntaps, tap = step % ntaps;
                for (int e = 0; e < 8; ++e) {
                    const int ci = chunk * 64 + kq * 8 + e;
                    o[e] = (_Float16)(n < Cout ? w[((size_t)n * Cin + ci) * ntaps + tap] : 0.f);
                }
            }
}

// the 7x7 stem (kernel comment): [4 steps][8 slots kq][CoutPad][8 halfs], e = (tap of the pair) * 4 + channel
static inline int conv_nsteps_stem_h() { return 4; }
static inline void conv_pack_weights_stem_h(const float* w, int Cout, ConvTile tile, _Float16* dst) {
    const ConvTileDims d = conv_tile_dims(tile);
    const int WN = d.BN / d.WGN, NT = WN / 32;
    const int CoutPad = conv_cout_pad(Cout, tile);
    for (int step = 0; step < 4; ++step)
        for (int kq = 0; kq < 8; ++kq)
            for (int slot = 0; slot < CoutPad; ++slot) {
                const int tn = slot / d.BN, within = slot % d.BN, wn = within / WN, w2 = within % WN;
                const int nt = w2 / 32, j = w2 % 32;
                const int n = tn * d.BN + wn * WN + j * NT + nt;
                _Float16* o = dst + (((size_t)step * 8 + kq) * CoutPad + slot) * 8;
                for (int e = 0; e < 8; ++e) {
                    const int ky = (kq >> 2) + 2 * step, kx = 2 * (kq & 3) + (e >> 2), c = e & 3;
                    o[e] = (_Float16)((n < Cout && c < 3 && ky < 7 && kx < 7) ? w[(((size_t)n * 3 + c) * 7 + ky) * 7 + kx] : 0.f);
                }
            }
}
// 64 output channels: the 128 x 64 block (weights packed for CT_128x64 / CT_128x64_DEEP)
static inline bool conv_stem_h_supports(ConvTile tile) { return tile == CT_128x64 || tile == CT_128x64_DEEP; }
static inline void conv_launch_stem_h(ConvArgs a, bool out16, hipStream_t s) {
    a.tiles_n = a.CoutPad / 64;
    const int grid = ((a.M + 127) / 128) * a.tiles_n;
    const int lds = ConvLdsH<128, 64>::BYTES;
    if (out16) TD_LAUNCH((k_conv_igemm_h<128, 64, 4, 1, 7, false, true, true>), dim3(grid), dim3(256), lds, s, a);
    else TD_LAUNCH((k_conv_igemm_h<128, 64, 4, 1, 7, false, false, true>), dim3(grid), dim3(256), lds, s, a);
}

template <int BM, int BN, int WGM, int WGN, bool IN16, bool OUT16>
static inline void conv_launch_h_t(const ConvArgs& a, int KS, hipStream_t s) {
    const int grid = ((a.M + BM - 1) / BM) * a.tiles_n;
    const int lds = ConvLdsH<BM, BN>::BYTES;
    if (KS == 3) TD_LAUNCH((k_conv_igemm_h<BM, BN, WGM, WGN, 3, IN16, OUT16>), dim3(grid), dim3(256), lds, s, a);
    else TD_LAUNCH((k_conv_igemm_h<BM, BN, WGM, WGN, 1, IN16, OUT16>), dim3(grid), dim3(256), lds, s, a);
}
template <bool IN16, bool OUT16>
static inline void conv_launch_h_io(const ConvArgs& a, const ConvTileDims& d, int KS, hipStream_t s) {
    if (d.BM == 128 && d.BN == 128) conv_launch_h_t<128, 128, 2, 2, IN16, OUT16>(a, KS, s);
    else if (d.BM == 64) conv_launch_h_t<64, 128, 2, 2, IN16, OUT16>(a, KS, s);
    else conv_launch_h_t<128, 64, 4, 1, IN16, OUT16>(a, KS, s);
}
// ng <= 3 convs of the same kernel form (tile, KS, in16, out16 -- the caller checks) in one launch
template <int BM, int BN, int WGM, int WGN, bool IN16, bool OUT16>
static inline void conv_launch_h_group_t(ConvGroupArgs& q, int KS0, hipStream_t s) {
    const int lds = ConvLdsH<BM, BN>::BYTES;
    if (KS0 == 3) TD_LAUNCH((k_conv_igemm_h_group<BM, BN, WGM, WGN, 3, 1, IN16, OUT16>), dim3(q.end[2]), dim3(256), lds, s, q);
    else TD_LAUNCH((k_conv_igemm_h_group<BM, BN, WGM, WGN, 1, 1, IN16, OUT16>), dim3(q.end[2]), dim3(256), lds, s, q);
}
// ng <= 3 convs of one kernel form (tile and storage types equal; 1x1 convs, or a 3x3 conv first and 1x1 convs behind it -- the caller checks)
static inline void conv_launch_h_group(const ConvArgs* a, int ng, ConvTile tile, int KS0, bool in16, bool out16, hipStream_t s) {
    const ConvTileDims d = conv_tile_dims(tile);
    ConvGroupArgs q;
    int blocks = 0;
    for (int g = 0; g < 3; ++g) {
        q.c[g] = a[g < ng ? g : ng - 1];
        q.c[g].tiles_n = q.c[g].CoutPad / d.BN;
        if (g < ng) blocks += ((q.c[g].M + d.BM - 1) / d.BM) * q.c[g].tiles_n;
        q.end[g] = blocks;
    }
    const bool io = in16 && out16;
    if (d.BM == 128 && d.BN == 128) { if (io) conv_launch_h_group_t<128, 128, 2, 2, true, true>(q, KS0, s); else conv_launch_h_group_t<128, 128, 2, 2, false, false>(q, KS0, s); }
    else if (d.BM == 64) { if (io) conv_launch_h_group_t<64, 128, 2, 2, true, true>(q, KS0, s); else conv_launch_h_group_t<64, 128, 2, 2, false, false>(q, KS0, s); }
    else { if (io) conv_launch_h_group_t<128, 64, 4, 1, true, true>(q, KS0, s); else conv_launch_h_group_t<128, 64, 4, 1, false, false>(q, KS0, s); }
}
// in16 / out16: storage type of the input (+ residual) / output map (see the header comment)
static inline void conv_launch_h(ConvArgs a, ConvTile tile, int KS, bool in16, bool out16, hipStream_t s) {
    const ConvTileDims d = conv_tile_dims(tile);
    a.tiles_n = a.CoutPad / d.BN;
    if (in16 && out16) conv_launch_h_io<true, true>(a, d, KS, s);
    else if (in16) conv_launch_h_io<true, false>(a, d, KS, s);
    else if (out16) conv_launch_h_io<false, true>(a, d, KS, s);
    else conv_launch_h_io<false, false>(a, d, KS, s);
}
