// td_conv.h -- NHWC implicit-GEMM convolution on the gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Covers every conv on the TDNet hot path: 3x3 with stride 1/2 and dilation
// 1..16 (resnet.py:32-37), 1x1 with stride 1/2/4 (downsample resnet.py:172-177, Encoding transformer.py:18-26,36),
// and the 7x7-s2 stem (resnet.py:132-133) / the 3x3-s2 first conv of the deep stem (:123) through the STEM variant.  BN is folded into weight/bias on the host, so the
// epilogue is  out = act(acc + bias[n] (+ residual[m][n])).
//
//   GEMM view:  M = Ho*Wo output pixels, N = Cout, K = taps * Cin.   D[m][n] = sum_k A[m][k] * B[k][n]
//   A[m][k] is gathered on the fly from the NHWC input (zero outside the image), B is the pre-packed weight.
//
// Data movement per block (256 threads = 4 waves, BM x BN output tile, BK = 32 reduction slice per step):
//   HBM/L2 -> registers (buffer_load_dwordx4; padded taps use an out-of-range offset, so the hardware bounds check
//   returns the zeros -- no branch) -> LDS (double buffered, one barrier per step) -> MFMA operands.
//   LDS images are [kq = 8 groups of 4 consecutive k][row][4 floats]: an MFMA lane (row = lane&31, half = lane>>5) reads
//   ONE ds_read_b128 per operand per 4 MFMAs, conflict free (consecutive rows = consecutive 16-B slots).  Inside a group
//   of 8 k the two lane-halves take k = 4*half + s, s = 0..3: a fixed permutation of the reduction order, applied to A
//   and B alike, so the sum is unchanged.
//   The K loop runs channel-chunk outer / tap inner so the 9 shifted re-reads of an input slab hit L2 back to back.
//   Output columns are permuted inside a wave (lane j owns NT consecutive channels j*NT..j*NT+NT-1) by the weight
//   packer, so the epilogue stores NT*4 contiguous bytes per lane instead of 4.
//
// Two software pipelines over the same tile code (template DEEP):
//   DEEP = false: tile s+1 is fetched while tile s is multiplied and written to LDS after the MFMAs (1 register set).
//   DEEP = true : tile s+2 is fetched while tile s is multiplied; tile s+1 (fetched a whole step earlier, so already
//                 landed) is written to LDS in between the MFMAs of step s (2 register sets, +32 VGPRs): no exposed
//                 vmcnt wait and no ds_write tail in front of the barrier.
#pragma once
#include "td_device.h"

#include <type_traits>

struct ConvArgs {
    const float* in;      // [H][W][Cin]            (STEM: [H][W][4], channel 3 = 0)
    const float* wp;      // packed weights [nsteps][8][CoutPad][4]
    const float* bias;    // [Cout]
    const float* resid;   // [M][Cout] or nullptr
    float* out;           // [M][Cout]
    int H, W, Cin;
    int Wo, Cout, CoutPad;
    int stride, dil, pad;
    int M;                // Ho*Wo
    int nsteps;           // (Cin/32)*KS*KS, STEM: ceil(KS*KS/8)
    int act;              // 0 none, 1 ReLU, 2 LeakyReLU(0.01)
    int tiles_n;          // CoutPad / BN
    int nbatch;           // > 1: batched 1x1 GEMMs (Winograd), see k_conv_igemm
};

// bijective "block b runs on XCD b%8" -> contiguous range of tiles per XCD (cdna_hip_programming.md T1)
TD_DEV int td_xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, q = bid >> 3, nq = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (nq + 1) : r * (nq + 1) + (xcd - r) * nq) + q;
}

TD_DEV f32x4 td_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
TD_DEV void td_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

template <int BM, int BN>
struct ConvLds {
    static constexpr int A_STRIDE = BM * 4 + 4;       // floats per kq group (+16 B pad: conflict-free ds_write_b128)
    static constexpr int B_STRIDE = BN * 4;
    static constexpr int A_FLOATS = 8 * A_STRIDE;
    static constexpr int B_FLOATS = 8 * B_STRIDE;
    static constexpr int BUF_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int BYTES = 2 * BUF_FLOATS * 4;
};

// ---- accumulator epilogue shared by the conv / GEMM kernels:  out[m][n] = act(acc + bias[n] (+ resid[m][n])) -----------
// acc[i][j][r] of a lane is row  m_base + 32 i + (r & 3) + 8 (r >> 2) + 4 half,  channel  n_base + NT l31 + j  (weight packer
// permutation).  Storing that directly is NT*4 bytes per lane and 16*MT*NT store instructions per wave, and the vector-memory
// path issues a store instruction only every few hundred cycles: on a K = 512 tile the stores cost as much as a quarter of the
// MFMA time.  With NT == 2 adjacent lanes swap half of a row pair (one DPP move per value): the even lane then holds 4
// consecutive channels of row r, the odd lane the same 4 channels of row r + 1, and a wave stores 16 bytes per lane, 256
// contiguous bytes per row, with a quarter of the store instructions.
// Activation without control flow: act is a kernel ARGUMENT, and `act == 1 ? .. : act == 2 ? ..` per element compiles to a chain
// of scalar compares and branches around every value (a dozen branches per 16-byte store, ~400 per tile and wave; a branch costs
// the wave tens of cycles during which it issues nothing).  max(v, 0) + slope * min(v, 0) with slope = 1 / 0 / 0.01 is the same
// function -- one of the two terms is always zero, so the result is rounded exactly like v, 0 or 0.01f * v.
// Non-finite values behave like the reference's modules: NaN propagates through every activation (a max/min formulation returns the
// OTHER operand for NaN and would hide a corrupted accumulator as 0), ReLU(-inf) = 0, LeakyReLU(-inf) = identity(-inf) = -inf.  The
// negative branch clamps to -FLT_MAX only when slope == 0, so that 0 * (-inf) cannot make a NaN; slope is wave-uniform, the clamp
// bound a loop-invariant SGPR.  Negative inputs of ReLU give -0.0f (0 * v), which compares and adds like +0.
TD_DEV float td_act_slope(int act) { return act == 1 ? 0.f : act == 2 ? 0.01f : 1.f; }
TD_DEV float td_activate(float v, float slope) {
    const float lo = slope == 0.f ? -3.402823466e+38f : -__builtin_inff();
    return v < 0.f ? slope * fmaxf(v, lo) : v;
}

// The 16-byte path of td_store_acc.  Everything here is VALU work of a wave that has no MFMA to issue, and while the other resident
// waves keep the matrix pipe busy such instructions issue slowly (tools/gemm_trace.hip: 6 us for an epilogue that takes 2 us on an
// idle CU), so it is kept short:
//  * addresses: one buffer descriptor over the output, a per-lane byte offset computed once, the row step added per store from an
//    SGPR (N is uniform); rows >= M and lanes with channel >= N fall outside the descriptor and the hardware drops the store -- no
//    64-bit address arithmetic, no predicate, no branch per store;
//  * ONE copy of the code (two copies selected by "has residual" make the compiler hoist the lane exchanges of all 8 MT row pairs
//    above the dispatch: 64 more live VGPRs, accumulator spills): the residual is read unconditionally through a descriptor with
//    ZERO records when there is none -- such loads return 0 without touching memory -- and the eight vectors of a 32-row group are
//    requested together before the first is used (one load, one wait, one store per row pair would expose the memory latency
//    sixteen times per tile; and behind a branch the compiler waits with vmcnt(0), which also waits for the previous STORE);
//  * NORES (compile time): no residual path at all;  PLAIN (compile time): no bias, no activation either (the Winograd GEMMs).
template <int MT, bool NORES, bool PLAIN>
TD_DEV void td_store_acc16(const f32x16 (&acc)[MT][2], float* out, const float* resid, int M, int N, float slope,
                           int m_base, int chan, bool cok, int half, int odd, f32x4 bv) {
    const unsigned bytes = (unsigned)M * (unsigned)N * 4u;
    const TdBuf out_buf = td_make_buf(out, bytes);
    const TdBuf res_buf = td_make_buf(resid, (NORES || !resid) ? 0u : bytes);
    const unsigned base = cok ? ((unsigned)(m_base + 4 * half + odd) * (unsigned)N + (unsigned)chan) * 4u : TD_BUF_OOB;
    const unsigned row_bytes = (unsigned)N * 4u;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        f32x4 rv[8];
        if (!NORES) {
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const int r = 2 * rp;
                rv[rp] = td_buf_ld4(res_buf, base + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes, 0u);
            }
            TD_SCHED_FENCE();
        }
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const int r = 2 * rp;
            const float a0 = acc[i][0][r], a1 = acc[i][1][r], c0 = acc[i][0][r + 1], c1 = acc[i][1][r + 1];
            const float x = td_swap1(odd ? a0 : c0), y = td_swap1(odd ? a1 : c1);   // even sends row r+1, odd sends row r
            f32x4 v;
            if (odd) { v[0] = x; v[1] = y; v[2] = c0; v[3] = c1; }
            else     { v[0] = a0; v[1] = a1; v[2] = x; v[3] = y; }
            if (!PLAIN) v = v + bv;
            if (!NORES) v = v + rv[rp];
            if (!PLAIN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = td_activate(v[e], slope);
            }
            td_buf_st4(out_buf, base + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes, v);
            TD_SCHED_FENCE();                                       // one row pair at a time (bounds the live registers)
        }
    }
}

// PLAIN: the caller guarantees bias == 0, act == 0, resid == nullptr (they are not looked at on the 16-byte path)
template <int MT, int NT, bool NORES = false, bool PLAIN = false>
TD_DEV void td_store_acc(const f32x16 (&acc)[MT][NT], float* out, const float* bias, const float* resid, int M, int N, int act,
                         int m_base, int n_base, int lane, const f32x4* bias_pre = nullptr) {
    // bias_pre: this lane's four bias values (channels n_base + 4 (l31 >> 1) ..), fetched by the caller long before the epilogue
    // (vmcnt is one in-order counter: waiting for a bias load issued here also waits for every older load of the wave).
    const int half = lane >> 5, l31 = lane & 31;
    const float slope = td_act_slope(act);
    auto activate = [&](float v) { return td_activate(v, slope); };
    if constexpr (NT == 2) {
        if ((N & 3) == 0 && ((((size_t)out) | ((size_t)resid)) & 15) == 0 && (size_t)(M + 128) * N < (1u << 29)) {   // wave-uniform
            const int odd = l31 & 1;
            const int chan = n_base + 4 * (l31 >> 1);
            const bool cok = chan < N;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (PLAIN) {}
            else if (bias_pre) bv = *bias_pre;
            else {
                // one unconditional (range-checked) load and its wait HERE: a load under `if (cok)` leaves the compiler unsure whether
                // it is pending, and it then waits with vmcnt(0) before every use -- i.e. for the previous store, sixteen times
                const bool al = (((size_t)bias) & 15) == 0;
                const TdBuf bias_buf = td_make_buf(bias, (unsigned)N * 4u);
                if (al) bv = td_buf_ld4(bias_buf, cok ? (unsigned)chan * 4u : TD_BUF_OOB, 0u);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = td_buf_ld1(bias_buf, cok ? (unsigned)(chan + e) * 4u : TD_BUF_OOB, 0u);
                }
                TD_PIN(bv);
            }
            td_store_acc16<MT, NORES || PLAIN, PLAIN>(acc, out, resid, M, N, slope, m_base, chan, cok, half, odd, bv);
            return;
        }
    }
    const int nb = n_base + l31 * NT;
    float bs[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bs[j] = (nb + j < N) ? bias[nb + j] : 0.f;
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
                float v = acc[i][j][r] + bs[j];
                if (resid) v += resid[o + j];
                out[o + j] = activate(v);
            }
        }
    }
}

// FLUSH (two-stage pipeline only; a power of two, 0 = off): every FLUSH K steps the MFMA accumulators are added into a second
// register set and cleared.  The MFMA sums its K products as ONE sequential fp32 chain per output (bit-identical to an fmaf chain);
// at K = 9 * 512 = 4608 (layer4 on the direct path) that chain's rounding error is ~sqrt(K) ulps of the partial sums and was
// 3.5x (rms) / 6.8x (max) what oneDNN's blocked summation leaves on the same graph with un-calibrated weights
// (tests/test_gpu_model.py::test_uncalibrated_reference_init).  Blocks of FLUSH * 32 = 512 products summed once bring the chain to
// sqrt(512) + sqrt(9): the direct kernel is then as accurate as the Winograd paths, whose GEMM chains are 512 long by construction.
// Cost: MT * NT * 16 more VGPRs and that many v_add every FLUSH steps; launched only for 3x3 convs with >= 64 K steps.
template <int BM, int BN, int WGM, int WGN, int KS, bool STEM, bool DEEP, int FLUSH = 0>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_conv_igemm(ConvArgs p) {
    static_assert(FLUSH == 0 || (DEEP && (FLUSH & (FLUSH - 1)) == 0 && FLUSH >= 2), "FLUSH: a power of two, two-stage pipeline only");
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int AL = BM / 32, BL = BN / 32;          // float4 slots per thread per step
    static_assert((AL == 2 || AL == 4) && (BL == 2 || BL == 4), "staging slots are spread over the 4 k-groups");
    using L = ConvLds<BM, BN>;
    constexpr int NTAPS = STEM ? 1 : KS * KS;
    TD_DYN_LDS(smem);
    float* lds = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WGN, wn = wave % WGN;

    // Batched launch (Winograd: 16 independent GEMMs, td_wino.h): the grid is nbatch consecutive groups of workgroups,
    // batch b reads in + b*M*Cin, the weights of batch b and writes out + b*M*Cout.  nbatch <= 1: plain conv.
    int bid_in_batch = blockIdx.x, nblk = gridDim.x;
    if (p.nbatch > 1) {
        nblk = gridDim.x / p.nbatch;
        const int b = blockIdx.x / nblk;
        bid_in_batch = blockIdx.x - b * nblk;
        p.in += (size_t)b * p.M * p.Cin;
        p.out += (size_t)b * p.M * p.Cout;
        p.wp += (size_t)b * p.nsteps * 8 * p.CoutPad * 4;
    }
    const int lin = td_xcd_remap(bid_in_batch, nblk);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread gather geometry: slot i -> row (tid>>3) + 32 i, k-group tid&7 ------------------------------
    const int a_row = tid >> 3, a_kq = tid & 7;
    int a_by[AL], a_bx[AL];
    unsigned a_off[AL];                                 // byte offset of the slot's top-left tap (may wrap; see below)
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + a_row + 32 * i;
        const int oy = m / p.Wo, ox = m - oy * p.Wo;
        a_by[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 28);
        a_bx[i] = ox * p.stride - p.pad;
        a_off[i] = (((unsigned)a_by[i] * (unsigned)p.W + (unsigned)a_bx[i]) * (unsigned)p.Cin + (STEM ? 0u : (unsigned)a_kq * 4u)) * 4u;
    }
    // Which of the KS x KS taps of slot i read inside the image: one bit per tap, computed ONCE.  The two bounds compares, the adds
    // and the logic per slot and K step were 40-60 VALU instructions per step -- on the issue port the MFMAs use, for 16-32 MFMAs.
    // The 128 x 64 tile (layer1, the stem) overlaps the side stream's attention chain in the frame.  Its loop needs 106 VGPRs, and at
    // 3 x 106 a SIMD has room for a fourth, side-stream wave: the kernel came out 3-4 % faster alone and 3-17 % SLOWER in the frame
    // (profiles/r02z_*).  Claiming 144 registers keeps the SIMDs to this kernel's three waves, as the older, fatter loop did by accident.
    if (BN == 64) TD_VGPR_FLOOR(143);
    unsigned a_taps[AL];
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
    const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 4u);
    const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
    unsigned b_off[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        b_off[i] = (unsigned)(kq * p.CoutPad + n0 + n) * 16u;
    }
    const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;

    // (chunk, tap) of the NEXT tile to fetch; advanced after every load_tile
    int l_step = 0, l_chunk = 0, l_tap = 0;
    auto load_tile = [&](f32x4 (&ra)[AL], f32x4 (&rb)[BL]) {
        int dy, dx;
        unsigned delta;                                 // byte offset of this tap/chunk relative to the slot's top-left tap
        bool tap_ok = true;
        if (STEM) {
            const int t = l_step * 8 + a_kq;            // one 4-channel pixel per k-group: 8 taps per step
            const int ky = t / KS;
            dy = ky; dx = t - ky * KS; tap_ok = t < KS * KS;
            delta = (unsigned)((dy * p.W + dx) * 4) * 4u;
        } else {
            const int ky = l_tap / KS;
            dy = ky * p.dil; dx = (l_tap - ky * KS) * p.dil;
            delta = (unsigned)((dy * p.W + dx) * p.Cin + l_chunk * 32) * 4u;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            bool ok;
            if (STEM) {
                const int iy = a_by[i] + dy, ix = a_bx[i] + dx;
                ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            } else ok = ((a_taps[i] >> l_tap) & 1u) != 0u;
            // valid taps: a_off + delta is the exact non-negative byte offset (mod 2^32 arithmetic); padded taps read zeros
            ra[i] = td_buf_ld4(in_buf, ok ? a_off[i] + delta : TD_BUF_OOB, 0u);
        }
        const unsigned wsoff = (unsigned)(l_step < p.nsteps ? l_step : p.nsteps - 1) * w_step_bytes;
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = td_buf_ld4(w_buf, b_off[i], wsoff);
        ++l_step;
        if (++l_tap == NTAPS) { l_tap = 0; ++l_chunk; }
    };
    // staging slot i of A / B -> LDS image of buffer `buf`
    auto store_a = [&](int buf, int i, const f32x4 (&ra)[AL]) {
        td_st4(lds + buf * L::BUF_FLOATS + a_kq * L::A_STRIDE + (a_row + 32 * i) * 4, ra[i]);
    };
    auto store_b = [&](int buf, int i, const f32x4 (&rb)[BL]) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        td_st4(lds + buf * L::BUF_FLOATS + L::A_FLOATS + kq * L::B_STRIDE + n * 4, rb[i]);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One K step: 16*MT*NT MFMAs on LDS buffer `buf`; operand fragments are fetched one k-group ahead of their MFMAs
    // (register double buffer).  With `st`, staging registers (sa, sb) are written to LDS buffer buf^1 in between.
    auto compute = [&](int buf, auto st, const f32x4 (&sa)[AL], const f32x4 (&sb)[BL]) {
        const float* As = lds + buf * L::BUF_FLOATS + (wm * WM + l31) * 4;
        const float* Bs = lds + buf * L::BUF_FLOATS + L::A_FLOATS + (wn * WN + l31) * 4;
        f32x4 af[2][MT], bf[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = td_ld4(As + half * L::A_STRIDE + i * 128);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[0][j] = td_ld4(Bs + half * L::B_STRIDE + j * 128);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[(g + 1) & 1][i] = td_ld4(As + (2 * g + 2 + half) * L::A_STRIDE + i * 128);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[(g + 1) & 1][j] = td_ld4(Bs + (2 * g + 2 + half) * L::B_STRIDE + j * 128);
            }
            if constexpr (decltype(st)::value) {
#pragma unroll
                for (int i = 0; i < AL; ++i) if (i * (4 / AL) == g) store_a(buf ^ 1, i, sa);
#pragma unroll
                for (int i = 0; i < BL; ++i) if (i * (4 / BL) == g) store_b(buf ^ 1, i, sb);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = td_mfma32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j]);
            // pin the interleave: LDS traffic of the NEXT group is spread between this group's MFMAs instead of being
            // sunk to its first use (where every group would start with an exposed LDS round trip)
            if (decltype(st)::value && MT == 2 && NT == 2 && AL == 4 && BL == 4) {
                TD_SCHED_GROUP(0x008, 2); TD_SCHED_GROUP(0x200, 1); TD_SCHED_GROUP(0x008, 2); TD_SCHED_GROUP(0x200, 1);
                if (g < 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { TD_SCHED_GROUP(0x100, 1); TD_SCHED_GROUP(0x008, 3); }
                } else {
                    TD_SCHED_GROUP(0x008, 12);
                }
            } else if (g < 3) {
#pragma unroll
                for (int r = 0; r < MT + NT; ++r) {
                    TD_SCHED_GROUP(0x100, 1);
                    TD_SCHED_GROUP(0x008, (4 * MT * NT) / (MT + NT));
                }
            } else {
                TD_SCHED_GROUP(0x008, 4 * MT * NT);
            }
        }
    };

    f32x4 ra[AL], rb[BL];
    if (!DEEP) {
        load_tile(ra, rb);
#pragma unroll
        for (int i = 0; i < AL; ++i) store_a(0, i, ra);
#pragma unroll
        for (int i = 0; i < BL; ++i) store_b(0, i, rb);
        __syncthreads();
        for (int step = 0; step < p.nsteps; ++step) {
            const bool more = step + 1 < p.nsteps;
            if (more) load_tile(ra, rb);                // global loads in flight while the MFMAs below run
            compute(step & 1, std::false_type{}, ra, rb);
            if (more) {
#pragma unroll
                for (int i = 0; i < AL; ++i) store_a((step + 1) & 1, i, ra);
#pragma unroll
                for (int i = 0; i < BL; ++i) store_b((step + 1) & 1, i, rb);
            }
            __syncthreads();
        }
    } else {
        // Branch-free: every step loads and stores unconditionally.  Past the last tile the weight step is clamped and
        // the activation offsets are either valid addresses or out of range (-> zeros); the surplus LDS image is never read.
        f32x4 ra2[AL], rb2[BL];
        load_tile(ra, rb);                              // tile 0
#pragma unroll
        for (int i = 0; i < AL; ++i) store_a(0, i, ra);
#pragma unroll
        for (int i = 0; i < BL; ++i) store_b(0, i, rb);
        load_tile(ra, rb);                              // tile 1 -> set 1, lands during step 0
        __syncthreads();
        // Whole two-step periods, then the odd last step on its own: with `if (last) break;` in the MIDDLE of the loop the compiler
        // keeps two copies of the accumulators and moves all of them (32 v_mov_b64 for a 128 x 128 tile) every iteration.
        f32x16 part[FLUSH ? MT : 1][FLUSH ? NT : 1];       // FLUSH: the blocks already summed
        if constexpr (FLUSH > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
        }
        int step = 0;
        for (; step + 1 < p.nsteps; step += 2) {
            load_tile(ra2, rb2);                        // even step: set 1 holds tile step+1, set 2 receives tile step+2
            compute(0, std::true_type{}, ra, rb);
            __syncthreads();
            load_tile(ra, rb);                          // odd step: set 2 holds tile step+2, set 1 receives tile step+3
            compute(1, std::true_type{}, ra2, rb2);
            __syncthreads();
            if constexpr (FLUSH > 0) {
                if (((step + 2) & (FLUSH - 1)) == 0) {  // wave-uniform
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            part[i][j] = part[i][j] + acc[i][j];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                        }
                }
            }
        }
        if (step < p.nsteps) compute(0, std::false_type{}, ra, rb);   // nsteps odd: the last tile is in LDS buffer 0, nothing left to stage
        if constexpr (FLUSH > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = part[i][j] + acc[i][j];
        }
    }

    td_store_acc<MT, NT>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * WM, n0 + wn * WN, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// host side: tile configuration, weight packing, launch
// ------------------------------------------------------------------------------------------------------------------
enum ConvTile { CT_128x128 = 0, CT_64x128 = 1, CT_128x64 = 2, CT_128x128_DEEP = 3, CT_64x128_DEEP = 4, CT_128x64_DEEP = 5,
                CT_COUNT = 6 };

struct ConvTileDims { int BM, BN, WGM, WGN; bool deep; };
static inline ConvTileDims conv_tile_dims(ConvTile t) {
    switch (t) {
        case CT_128x128: return {128, 128, 2, 2, false};
        case CT_64x128: return {64, 128, 2, 2, false};
        case CT_128x64: return {128, 64, 4, 1, false};
        case CT_128x128_DEEP: return {128, 128, 2, 2, true};
        case CT_64x128_DEEP: return {64, 128, 2, 2, true};
        default: return {128, 64, 4, 1, true};
    }
}

// `deep` (tdnet_opts.pipeline): 0 = single-stage prefetch, 1 = two-stage pipeline (the default: measured +4..6 % on every layer
// shape on MI355X, profiles/r01b_kernel_probe.txt).
// Choose the tile: 128-wide N when Cout allows, and the smaller M tile when 128x128 would leave CUs idle.
// Cost model: the launch ends when the busiest CU has finished its share of the workgroups, so what matters is the
// per-CU quantisation ceil(blocks / 256 CUs) times the work of one block, divided by the tile's measured relative
// throughput (profiles/r01b_kernel_probe.txt: 128x128 1.00, 64x128 0.97, 128x64 0.95).  At 1024x2048 this picks 128x128
// for layers 3-4 (1024 blocks = 4 per CU); at the native 769x1537 (147 M-tiles -> 588 blocks = 2.3 per CU) it picks
// 64x128 (1172 blocks = 4.6 per CU), which removes most of the last-round idle time.
static inline ConvTile conv_pick_tile(int M, int Cout, bool deep) {
    ConvTile best = CT_128x64;
    if (Cout > 64) {
        static const struct { ConvTile t; int bm, bn; double eff; } cand[3] = {
            {CT_128x128, 128, 128, 1.00}, {CT_64x128, 64, 128, 0.97}, {CT_128x64, 128, 64, 0.95}};
        double best_cost = 0.0;
        for (int i = 0; i < 3; ++i) {
            const long blocks = (long)((M + cand[i].bm - 1) / cand[i].bm) * ((Cout + cand[i].bn - 1) / cand[i].bn);
            const double occ = blocks >= 512 ? 1.0 : 0.8;      // fewer than 2 workgroups per CU: no co-resident wave to hide stalls
            const double cost = (double)((blocks + 255) / 256) * cand[i].bm * cand[i].bn / (cand[i].eff * occ);
            if (i == 0 || cost < best_cost) { best_cost = cost; best = cand[i].t; }
        }
    }
    return deep ? (ConvTile)(best + 3) : best;
}

static inline int conv_cout_pad(int Cout, ConvTile t) {
    const int BN = conv_tile_dims(t).BN;
    return (Cout + BN - 1) / BN * BN;
}
static inline int conv_nsteps(int Cin, int KS, int stem) { return stem == 2 ? KS : stem ? (KS * KS + 7) / 8 : (Cin / 32) * KS * KS; }

// Pack BN-folded OIHW weights into the LDS image order of `tile`: [step][kq][slot][4], where packed column `slot`
// holds output channel  tile_n*BN + wn*WN + j*NT + nt   for  slot = tile_n*BN + wn*WN + nt*32 + j.
// stem = 2: the 7x7 stem on the PACKED-ROW image (td_conv_ad.h STEM = 2): step = kernel row ky, k = 4 kq + e = 3 kx + c for k < 21.
static inline void conv_pack_weights(const float* w, int Cout, int Cin, int KS, int stem, ConvTile tile, float* dst) {
    const ConvTileDims d = conv_tile_dims(tile);
    const int WN = d.BN / d.WGN, NT = WN / 32;
    const int CoutPad = conv_cout_pad(Cout, tile);
    const int nsteps = conv_nsteps(Cin, KS, stem), ntaps = KS * KS;
    for (int step = 0; step < nsteps; ++step)
        for (int kq = 0; kq < 8; ++kq)
            for (int slot = 0; slot < CoutPad; ++slot) {
                const int tn = slot / d.BN, within = slot % d.BN, wn = within / WN, w2 = within % WN;
                const int nt = w2 / 32, j = w2 % 32;
                const int n = tn * d.BN + wn * WN + j * NT + nt;
                float* o = dst + (((size_t)step * 8 + kq) * CoutPad + slot) * 4;
                for (int e = 0; e < 4; ++e) {
                    float v = 0.f;
                    if (n < Cout) {
                        if (stem == 2) {
                            const int k = kq * 4 + e;                  // the row's 21 (kx, c) products, then three zero columns, then two zero k-groups
                            if (k < 3 * KS) v = w[((size_t)n * 3 + k % 3) * ntaps + step * KS + k / 3];
                        } else if (stem) {
                            const int t = step * 8 + kq;
                            if (t < ntaps && e < 3) v = w[((size_t)n * 3 + e) * ntaps + t];
                        } else {
                            const int chunk = step / ntaps, tap = step % ntaps, ci = chunk * 32 + kq * 4 + e;
                            v = w[((size_t)n * Cin + ci) * ntaps + tap];
                        }
                    }
                    o[e] = v;
                }
            }
}

template <int BM, int BN, int WGM, int WGN, bool DEEP>
static inline void conv_launch_t(const ConvArgs& a, int KS, bool stem, hipStream_t s) {
    const int grid = ((a.M + BM - 1) / BM) * a.tiles_n * (a.nbatch > 1 ? a.nbatch : 1);
    const int lds = ConvLds<BM, BN>::BYTES;
    if (stem && KS == 7) TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 7, true, DEEP>), dim3(grid), dim3(256), lds, s, a);
    else if (stem) TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 3, true, DEEP>), dim3(grid), dim3(256), lds, s, a);
    else if (KS == 3 && DEEP && a.nsteps >= 64) TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 3, false, DEEP, DEEP ? 16 : 0>), dim3(grid), dim3(256), lds, s, a);   // long K: blocked summation
    else if (KS == 3) TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 3, false, DEEP>), dim3(grid), dim3(256), lds, s, a);
    else TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 1, false, DEEP>), dim3(grid), dim3(256), lds, s, a);
}

static inline void conv_launch(ConvArgs a, ConvTile tile, int KS, bool stem, hipStream_t s) {
    const ConvTileDims d = conv_tile_dims(tile);
    a.tiles_n = a.CoutPad / d.BN;
    switch (tile) {
        case CT_128x128: conv_launch_t<128, 128, 2, 2, false>(a, KS, stem, s); break;
        case CT_64x128: conv_launch_t<64, 128, 2, 2, false>(a, KS, stem, s); break;
        case CT_128x64: conv_launch_t<128, 64, 4, 1, false>(a, KS, stem, s); break;
        case CT_128x128_DEEP: conv_launch_t<128, 128, 2, 2, true>(a, KS, stem, s); break;
        case CT_64x128_DEEP: conv_launch_t<64, 128, 2, 2, true>(a, KS, stem, s); break;
        default: conv_launch_t<128, 64, 4, 1, true>(a, KS, stem, s); break;
    }
}
