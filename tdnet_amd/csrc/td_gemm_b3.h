// td_gemm_b3.h -- fp32-ACCURATE batched GEMM on the bf16 MFMA (tdnet_opts.precision = 2, opt-in; the default stays the exact-fp32 MFMA).
//
//   out[b][m][n] = sum_k A[b][m][k] W[b][k][n]          (ROLE 1: the 36 GEMMs of a Winograd F(4x4) conv; ROLE 0 / 2: a stride-1 1x1 conv)
//
// gfx950 has no TF32-like MFMA, and its fp32 MFMA runs at the VECTOR rate (157 TF), 1/16 of the bf16 MFMA.  An fp32 value is the exact sum of
// three bf16 values, x = x0 + x1 + x2 (|x1| <= 2^-9 |x|, |x2| <= 2^-18 |x|: 8 + 8 + 8 significand bits and a sign each, round to nearest),
// so a product is a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + terms below 2^-26 |a b|: SIX bf16 MFMAs with fp32 accumulation carry
// every product to fp32 accuracy at 16 / 6 = 2.67x the fp32 MFMA's rate.  The sums are NOT bit-identical to the fp32 MFMA's (the matrix
// core adds the 16 products of an instruction in its own order) -- the mode is held to the same gates as the fp32 kernels against fp64.
//
// Operands: W is split ON THE HOST, once (gemm_b3_pack: [K/16][part 3][k-half 2][NPad][8 bf16], 6 bytes per weight); A stays fp32 in HBM (the
// Winograd input transform and the activation maps are unchanged, 4 bytes per element) and is split BY THE MATRIX WAVES, in registers,
// one K step ahead of its MFMAs: 44 VALU instructions per 32 x 16 fragment against 24 MFMAs of 32 cycles -- a wave tile is 64 rows x ALL 128
// columns of the workgroup tile so that a fragment is split once, not once per wave column.
//
// Tile (64 WR) x 128, K step 16, WR matrix waves + WR LOADER waves (td_conv_hd.h k_conv_dma_h3p: a `buffer_load .. lds` holds its wave for
// 100+ cycles, so the matrix waves never issue one), a ring of NBUF LDS buffers filled by LDS-DMA NBUF - 1 steps ahead, one bare barrier per
// step, persistent XCD-aware tile lists as in td_gemm.h.
//   A image: [row][4 slots of 16 B] (64 bytes = the step's 16 floats), slot XOR-swizzled with (row >> 2) & 3 on the SOURCE address: the 16
//            lanes of a ds_read_b128 phase (consecutive rows) cover the 16 slots of a 256-byte bank row;
//   B image: [part][k-half][128 columns][16 B], a linear copy of the packed weights; column permutation of conv_pack_weights (BN = 128, two
//            64-column groups of NT = 2), so the epilogue is td_store_acc's 16-byte path, once per column group.
#pragma once
#include "td_gemm.h"

#include <cstring>
#include <vector>

template <int WR>
struct GemmB3Geom {
    static constexpr int BM = 64 * WR, BN = 128, NL = WR;              // matrix waves = loader waves = WR
    static constexpr int A_BYTES = BM * 64, B_BYTES = 12 * 1024, BUF_BYTES = A_BYTES + B_BYTES;
    static constexpr int NBUF = WR == 4 ? 5 : 3, LA = NBUF - 1;         // ring depth, steps of lookahead of the loaders
    static constexpr int LDS_BYTES = NBUF * BUF_BYTES;
    static constexpr int APL = 4, BPL = 12 / WR, PPS = APL + BPL;       // pieces per loader and step
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// eight consecutive fp32 of a row (two 16-byte slots) -> their three bf16 parts, packed two per dword (td_mfma32_bf16 operands)
TD_DEV void td_split3(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
        const unsigned ph = td_pk_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);   // exact
        const unsigned pm = td_pk_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);   // exact
        h[i] = ph; m[i] = pm; l[i] = td_pk_bf16(sa, sb);
    }
}

// VAR (compile time, A/B of the schedule; the shipped value is GEMM_B3_VAR): bit 0 = the first B fragments of step g + 1 are read before the
// barrier that ends step g (no LDS round trip in front of a step's first MFMAs), bit 1 = the split of the next step's A is pinned between
// the MFMAs (two VALU instructions per MFMA) instead of where the scheduler sinks it, bit 2 = single (unpacked) v_sub_f32 in the split,
// bit 3 = the whole step as twelve fenced groups in an explicit order (with bit 0's prefetch).
// SKIP (compile time; != 0 only in -DTD_B3_PROBE builds, tools/b3_probe.py --skip: timing probes, the results are garbage): 1 = no MFMAs,
// 2 = no LDS-DMA, 4 = no split, 8 = no fragment reads of B.
#ifndef GEMM_B3_VAR
#define GEMM_B3_VAR 0
#endif
#define TD_B3_SKIP(bit) ((SKIP & (bit)) != 0)
template <int VAR>
TD_DEV void td_split3v(const f32x4& x0, const f32x4& x1, u32x4& h, u32x4& m, u32x4& l) {
    if constexpr ((VAR & 4) == 0) { td_split3(x0, x1, h, m, l); return; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
        const unsigned ph = td_pk_bf16(a, b);
        const float ra = td_sub1(a, __builtin_bit_cast(float, ph << 16)), rb = td_sub1(b, __builtin_bit_cast(float, ph & 0xffff0000u));
        const unsigned pm = td_pk_bf16(ra, rb);
        const float sa = td_sub1(ra, __builtin_bit_cast(float, pm << 16)), sb = td_sub1(rb, __builtin_bit_cast(float, pm & 0xffff0000u));
        h[i] = ph; m[i] = pm; l[i] = td_pk_bf16(sa, sb);
    }
}

// one pair of fp32 -> one dword of each part (11 VALU: 3 cvt_pk, 4 shift / and, 4 sub)
TD_DEV void td_split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = td_pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = td_pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    l = td_pk_bf16(sa, sb);
}

template <int WR, int ROLE, int VAR, int SKIP = 0>
TD_KERNEL void TD_LAUNCH_BOUNDS(128 * WR, WR == 4 ? 1 : 2) k_gemm_b3(GemmArgs p) {
    using G = GemmB3Geom<WR>;
    TD_DYN_LDS(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int nsteps = p.K >> 4;

    // ---- this workgroup's tile list (td_gemm.h): range of XCD (bid % 8), positions q, q + G8, ... -------------------------------
    const int per_batch = p.tiles_m * p.tiles_n, total = per_batch * p.nbatch;
    const int NX = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % NX, q = blockIdx.x / NX;
    const int G8 = ((int)gridDim.x + NX - 1 - xcd) / NX;
    const int nq = total / NX, rem = total % NX;
    const int xbase = xcd < rem ? xcd * (nq + 1) : rem * (nq + 1) + (xcd - rem) * nq;
    const int xcount = nq + (xcd < rem ? 1 : 0);
    const int my_tiles = q < xcount ? (xcount - q + G8 - 1) / G8 : 0;
    if (my_tiles == 0) return;
    struct TilePos { int b, tm, tn; };
    const int lin0 = xbase + q, r00 = lin0 % per_batch;
    const TilePos pos0 = {lin0 / per_batch, r00 / p.tiles_n, r00 % p.tiles_n};
    const int dB = G8 / per_batch, dR = G8 % per_batch, dTm = dR / p.tiles_n, dTn = dR % p.tiles_n;
    auto advance = [&](TilePos& t) {
        t.tn += dTn;
        const int c = t.tn >= p.tiles_n ? 1 : 0;
        t.tn -= c ? p.tiles_n : 0;
        t.tm += dTm + c;
        const int c2 = t.tm >= p.tiles_m ? 1 : 0;
        t.tm -= c2 ? p.tiles_m : 0;
        t.b += dB + c2;
    };
    const int gsteps = my_tiles * nsteps;                              // K steps of this workgroup over all its tiles

    if (wave >= WR) {
        // =========================== loader wave pw: A pieces pw + NL j (16 rows each), B pieces pw + NL jb ===========================
        const int pw = wave - WR;
        const unsigned w_step_bytes = 6u * (unsigned)p.NPad * 16u;
        const unsigned a_bytes = (unsigned)p.M * (unsigned)p.K * 4u, w_bytes = (unsigned)nsteps * w_step_bytes;
        int l_tile = 0, l_step = 0;
        TilePos lpos = pos0;
        TdBuf a_buf, w_buf;
        unsigned a_off[G::APL], b_off[G::BPL];
        auto enter_tile = [&]() {
            a_buf = td_make_buf(p.a + (size_t)lpos.b * p.MP * p.K, a_bytes);
            w_buf = td_make_buf(reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wp) + (p.wshare ? (size_t)0 : (size_t)lpos.b * w_bytes)), w_bytes);
#pragma unroll
            for (int j = 0; j < G::APL; ++j) {
                const int row = 16 * (pw + G::NL * j) + (lane >> 2);
                const int m = lpos.tm * G::BM + row;
                const int sl = (lane & 3) ^ ((row >> 2) & 3);
                a_off[j] = m < p.M ? ((unsigned)m * (unsigned)p.K + (unsigned)sl * 4u) * 4u : TD_BUF_OOB;
            }
#pragma unroll
            for (int jb = 0; jb < G::BPL; ++jb) {
                const int pb = pw + G::NL * jb;                        // piece pb = 2 (part, k-half) + column half
                b_off[jb] = (unsigned)((pb >> 1) * p.NPad + lpos.tn * G::BN + (pb & 1) * 64 + lane) * 16u;
            }
        };
        int ibuf = 0;                                                  // ring position of the next step to issue
        auto issue = [&]() {
            char* base = smem + ibuf * G::BUF_BYTES;
            const bool live = l_tile < my_tiles && !TD_B3_SKIP(2);     // past the end: zero-fill pieces keep the counted waits uniform
#pragma unroll
            for (int j = 0; j < G::APL; ++j)
                td_buf_ld16_lds(a_buf, base + (pw + G::NL * j) * 1024, (live && !TD_B3_SKIP(16)) ? a_off[j] : TD_BUF_OOB, live ? (unsigned)l_step * 64u : 0u);
#pragma unroll
            for (int jb = 0; jb < G::BPL; ++jb)
                td_buf_ld16_lds(w_buf, base + G::A_BYTES + (pw + G::NL * jb) * 1024, (live && !TD_B3_SKIP(32)) ? b_off[jb] : TD_BUF_OOB, live ? (unsigned)l_step * w_step_bytes : 0u);
            ibuf = ibuf + 1 == G::NBUF ? 0 : ibuf + 1;
            if (l_tile < my_tiles && ++l_step == nsteps) {
                l_step = 0;
                if (++l_tile < my_tiles) { advance(lpos); enter_tile(); }
            }
        };
        enter_tile();
#pragma unroll
        for (int s = 0; s < G::LA; ++s) issue();                       // steps 0 .. LA - 1
        TD_WAIT_VM_PIECES((G::LA - 2) * G::PPS);                       // steps 0 and 1 have landed
        TD_BARRIER_RAW();
        for (int g = 0; g < gsteps; ++g) {
            issue();                                                   // step g + LA into the buffer step g - 1 left
            TD_WAIT_VM_PIECES((G::LA - 2) * G::PPS);                   // steps <= g + 2 have landed (the matrix waves read one step ahead)
            TD_BARRIER_RAW();
        }
        TD_WAIT_VM_PIECES(0);                                          // surplus pieces must not land in an LDS that has been handed on
        return;
    }

    // =========================================== matrix wave: 64 rows x 128 columns ===========================================
    const int half = lane >> 5, l31 = lane & 31;
    unsigned a_rd[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 64 + i * 32 + l31;
#pragma unroll
        for (int e = 0; e < 2; ++e) a_rd[i][e] = (unsigned)(row * 64 + (((2 * half + e) ^ ((row >> 2) & 3)) << 4));
    }
    const unsigned b_rd = (unsigned)(G::A_BYTES + (half * 128 + l31) * 16);

    f32x16 acc[2][2][2];                                               // [column group][row block][column block]
    auto zero_acc = [&]() {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[g2][i][j][r] = 0.f;
    };
    u32x4 ah[2], am[2], al[2];                                         // the current step's A, split
    u32x4 bpre[2];                                                     // VAR bit 0: part 0 of the current step's first column group
    auto load_a = [&](int buf, f32x4 (&x)[2][2]) {
        const char* base = smem + buf * G::BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) x[i][e] = *reinterpret_cast<const f32x4*>(base + a_rd[i][e]);
    };
    auto load_b = [&](int buf, int part, int g2, int j) {
        return *reinterpret_cast<const u32x4*>(smem + buf * G::BUF_BYTES + b_rd + part * 4096 + g2 * 1024 + j * 512);
    };
    TilePos spos = pos0;
    zero_acc();
    TD_BARRIER_RAW();                                                  // the loaders' prologue: steps 0 and 1 are in LDS
    {
        f32x4 x[2][2];
        load_a(0, x);
#pragma unroll
        for (int i = 0; i < 2; ++i) td_split3v<VAR>(x[i][0], x[i][1], ah[i], am[i], al[i]);
        if constexpr (VAR & 9) { bpre[0] = load_b(0, 0, 0, 0); bpre[1] = load_b(0, 0, 0, 1); }
    }
    int cb = 0;                                                        // ring position of the current step
    for (int t = 0; t < my_tiles; ++t) {
        for (int st = 0; st < nsteps; ++st) {
            const int nb = cb + 1 == G::NBUF ? 0 : cb + 1;
            if constexpr ((VAR & 8) != 0 && SKIP == 0) {
                // EXPLICIT schedule (VAR bit 3): twelve groups of four MFMAs (one product x one column group each), fenced; the split of the next
                // step's A rides in eight of them (one pair chain = 11 VALU per group), the fragment reads two groups ahead of their use
                f32x4 nx[2][2];
                load_a(nb, nx);
                u32x4 bm0[2], bl0[2], bh1[2], bm1[2], bl1[2], nh[2], nm[2], nl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) { bm0[j] = load_b(cb, 1, 0, j); bl0[j] = load_b(cb, 2, 0, j); }
                auto mm = [&](const u32x4 (&A)[2], const u32x4 (&B)[2], int g2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(A[i], B[j], acc[g2][i][j]);
                };
                auto chunk = [&](int c) {                                  // c = 0..7: row block c >> 2, pair c & 3 of its eight values
                    const int i = c >> 2, q = c & 3;
                    const float a = q < 2 ? nx[i][0][2 * q] : nx[i][1][2 * q - 4], b = q < 2 ? nx[i][0][2 * q + 1] : nx[i][1][2 * q - 3];
                    unsigned h_, m_, l_;
                    td_split3_pair(a, b, h_, m_, l_);
                    nh[i][q] = h_; nm[i][q] = m_; nl[i][q] = l_;
                };
                TD_SCHED_FENCE();
                mm(ah, bpre, 0);
                TD_SCHED_FENCE();
                mm(am, bpre, 0); chunk(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) bh1[j] = load_b(cb, 0, 1, j);
                TD_SCHED_FENCE();
                mm(al, bpre, 0); chunk(1);
                TD_SCHED_FENCE();
                mm(ah, bm0, 0); chunk(2);
#pragma unroll
                for (int j = 0; j < 2; ++j) bm1[j] = load_b(cb, 1, 1, j);
                TD_SCHED_FENCE();
                mm(am, bm0, 0); chunk(3);
                TD_SCHED_FENCE();
                mm(ah, bl0, 0); chunk(4);
#pragma unroll
                for (int j = 0; j < 2; ++j) bl1[j] = load_b(cb, 2, 1, j);
                TD_SCHED_FENCE();
                mm(ah, bh1, 1); chunk(5);
                TD_SCHED_FENCE();
                mm(am, bh1, 1); chunk(6);
                TD_SCHED_FENCE();
                mm(al, bh1, 1); chunk(7);
                TD_SCHED_FENCE();
                mm(ah, bm1, 1);
#pragma unroll
                for (int j = 0; j < 2; ++j) bpre[j] = load_b(nb, 0, 0, j);
                TD_SCHED_FENCE();
                mm(am, bm1, 1);
                TD_SCHED_FENCE();
                mm(ah, bl1, 1);
                TD_SCHED_FENCE();
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[i] = nh[i]; am[i] = nm[i]; al[i] = nl[i]; }
            } else {
            f32x4 nx[2][2];
            if constexpr (!TD_B3_SKIP(4)) load_a(nb, nx);                        // the NEXT step's A (landed: the loaders run two steps ahead of the barrier)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                u32x4 bh[2], bm[2], bl[2];
                if constexpr (!TD_B3_SKIP(8)) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bh[j] = ((VAR & 1) && g2 == 0) ? bpre[j] : load_b(cb, 0, g2, j);
                        bm[j] = load_b(cb, 1, g2, j);
                        bl[j] = load_b(cb, 2, g2, j);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) { bh[j] = ah[j]; bm[j] = am[j]; bl[j] = al[j]; }
                }
                if constexpr (TD_B3_SKIP(1)) {                         // no MFMAs: the fragments must still be read and split
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[g2][0][j][0] += __builtin_bit_cast(float, bh[j][0] ^ bm[j][1] ^ bl[j][2] ^ ah[j][0] ^ am[j][1] ^ al[j][2]);
                }
                if constexpr (!TD_B3_SKIP(1)) {
                    // six products per (row block, column block); four independent accumulators between two uses of one.  Part 0 of B first
                    // (VAR bit 0: its registers are the ones refilled for the next step)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(ah[i], bh[j], acc[g2][i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(am[i], bh[j], acc[g2][i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(al[i], bh[j], acc[g2][i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(ah[i], bm[j], acc[g2][i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(am[i], bm[j], acc[g2][i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(ah[i], bl[j], acc[g2][i][j]);
                }
            }
            if constexpr (VAR & 1) { bpre[0] = load_b(nb, 0, 0, 0); bpre[1] = load_b(nb, 0, 0, 1); }
            if constexpr (!TD_B3_SKIP(4)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) td_split3v<VAR>(nx[i][0], nx[i][1], ah[i], am[i], al[i]);
            }
            if constexpr (VAR & 2) {
                // the step's instruction mix: 48 MFMAs, ~92 VALU of the split, 16 LDS reads.  LDS reads first (they feed everything), then
                // two VALU per MFMA: the split of the next step's A rides in the issue slots the 32-cycle MFMAs leave free
                TD_SCHED_GROUP(0x100, 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) { TD_SCHED_GROUP(0x008, 1); TD_SCHED_GROUP(0x100, 2); }
#pragma unroll
                for (int r = 0; r < 44; ++r) { TD_SCHED_GROUP(0x008, 1); TD_SCHED_GROUP(0x002, 2); }
            }
            }
            TD_BARRIER_RAW();
            cb = nb;
        }
        float* outb = p.out + (size_t)spos.b * p.MP * p.N;
        if constexpr (TD_B3_SKIP(64)) {                                // no epilogue stores (keeps the accumulators live through one value)
            float sum = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += acc[g2][i][j][r];
            if (sum == 1.2345f) outb[lane] = sum;
        } else
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
            td_store_acc<2, 2, ROLE != 0, ROLE == 1>(acc[g2], outb, p.bias, p.resid, p.M, p.N, ROLE == 1 ? 0 : p.act, spos.tm * G::BM + wave * 64,
                                                     spos.tn * G::BN + g2 * 64, lane);
        zero_acc();
        advance(spos);
    }
}

// ---- the same GEMM WITHOUT loader waves: 256 x 128 tile, four matrix waves that issue their own LDS-DMA between MFMA groups (td_gemm_dma.h),
// TWO workgroups per CU -- the second workgroup's MFMAs cover this one's fragment reads, splits, DMA issue, barrier skew and, above all, its
// epilogue (a quarter of a K = 512 tile's time when nothing runs beside it: profiles/r06d_*).  Two LDS buffers per operand; A runs one step
// ahead of B (it is split one step ahead): step g issues A(g + 2) and B(g + 1), reads A(g + 1) and B(g), and ends with vmcnt(0) + barrier.
struct GemmB3mGeom {
    static constexpr int BM = 256, BN = 128;
    static constexpr int A_BYTES = BM * 64, B_BYTES = 12 * 1024, LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;
};
template <int ROLE>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_gemm_b3m(GemmArgs p) {
    using G = GemmB3mGeom;
    TD_DYN_LDS(smem);
    char* const bbase = smem + 2 * G::A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int nsteps = p.K >> 4;
    const int per_batch = p.tiles_m * p.tiles_n, total = per_batch * p.nbatch;
    const int NX = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % NX, q = blockIdx.x / NX;
    const int G8 = ((int)gridDim.x + NX - 1 - xcd) / NX;
    const int nq = total / NX, rem = total % NX;
    const int xbase = xcd < rem ? xcd * (nq + 1) : rem * (nq + 1) + (xcd - rem) * nq;
    const int xcount = nq + (xcd < rem ? 1 : 0);
    const int my_tiles = q < xcount ? (xcount - q + G8 - 1) / G8 : 0;
    if (my_tiles == 0) return;
    struct TilePos { int b, tm, tn; };
    const int lin0 = xbase + q, r00 = lin0 % per_batch;
    const TilePos pos0 = {lin0 / per_batch, r00 / p.tiles_n, r00 % p.tiles_n};
    const int dB = G8 / per_batch, dR = G8 % per_batch, dTm = dR / p.tiles_n, dTn = dR % p.tiles_n;
    auto advance = [&](TilePos& t) {
        t.tn += dTn;
        const int c = t.tn >= p.tiles_n ? 1 : 0;
        t.tn -= c ? p.tiles_n : 0;
        t.tm += dTm + c;
        const int c2 = t.tm >= p.tiles_m ? 1 : 0;
        t.tm -= c2 ? p.tiles_m : 0;
        t.b += dB + c2;
    };
    const unsigned w_step_bytes = 6u * (unsigned)p.NPad * 16u;
    const unsigned a_bytes = (unsigned)p.M * (unsigned)p.K * 4u, w_bytes = (unsigned)nsteps * w_step_bytes;

    // ---- two loader cursors: A (two steps ahead of the MFMAs) and B (one step ahead); past the last tile both issue zero-fill pieces -------
    struct Cur { int tile, step; TilePos pos; };
    Cur ca = {0, 0, pos0}, cw = {0, 0, pos0};
    TdBuf a_buf, w_buf;
    unsigned a_off[4], b_off[3];
    auto enter_a = [&]() {
        a_buf = td_make_buf(p.a + (size_t)ca.pos.b * p.MP * p.K, a_bytes);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 16 * (wave + 4 * j) + (lane >> 2);
            const int m = ca.pos.tm * G::BM + row;
            const int sl = (lane & 3) ^ ((row >> 2) & 3);
            a_off[j] = m < p.M ? ((unsigned)m * (unsigned)p.K + (unsigned)sl * 4u) * 4u : TD_BUF_OOB;
        }
    };
    auto enter_w = [&]() {
        w_buf = td_make_buf(reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.wp) + (p.wshare ? (size_t)0 : (size_t)cw.pos.b * w_bytes)), w_bytes);
#pragma unroll
        for (int jb = 0; jb < 3; ++jb) {
            const int pb = wave + 4 * jb;
            b_off[jb] = (unsigned)((pb >> 1) * p.NPad + cw.pos.tn * G::BN + (pb & 1) * 64 + lane) * 16u;
        }
    };
    int abuf_i = 0, wbuf_i = 0;                                        // buffer the next A / B step goes to
    auto issue_a = [&](int j) {
        const bool live = ca.tile < my_tiles;
        td_buf_ld16_lds(a_buf, smem + abuf_i * G::A_BYTES + (wave + 4 * j) * 1024, live ? a_off[j] : TD_BUF_OOB, live ? (unsigned)ca.step * 64u : 0u);
    };
    auto end_a = [&]() {
        abuf_i ^= 1;
        if (ca.tile < my_tiles && ++ca.step == nsteps) { ca.step = 0; if (++ca.tile < my_tiles) { advance(ca.pos); enter_a(); } }
    };
    auto issue_w = [&](int jb) {
        const bool live = cw.tile < my_tiles;
        td_buf_ld16_lds(w_buf, bbase + wbuf_i * G::B_BYTES + (wave + 4 * jb) * 1024, live ? b_off[jb] : TD_BUF_OOB, live ? (unsigned)cw.step * w_step_bytes : 0u);
    };
    auto end_w = [&]() {
        wbuf_i ^= 1;
        if (cw.tile < my_tiles && ++cw.step == nsteps) { cw.step = 0; if (++cw.tile < my_tiles) { advance(cw.pos); enter_w(); } }
    };

    const int half = lane >> 5, l31 = lane & 31;
    unsigned a_rd[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 64 + i * 32 + l31;
#pragma unroll
        for (int e = 0; e < 2; ++e) a_rd[i][e] = (unsigned)(row * 64 + (((2 * half + e) ^ ((row >> 2) & 3)) << 4));
    }
    const unsigned b_rd = (unsigned)((half * 128 + l31) * 16);
    f32x16 acc[2][2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[g2][i][j][r] = 0.f;
    };
    u32x4 ah[2], am[2], al[2];
    auto load_a = [&](int buf, f32x4 (&x)[2][2]) {
        const char* base = smem + buf * G::A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) x[i][e] = *reinterpret_cast<const f32x4*>(base + a_rd[i][e]);
    };
    auto load_b = [&](int buf, int part, int g2, int j) {
        return *reinterpret_cast<const u32x4*>(bbase + buf * G::B_BYTES + b_rd + part * 4096 + g2 * 1024 + j * 512);
    };

    TilePos spos = pos0;
    zero_acc();
    enter_a(); enter_w();
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_a(j);
    end_a();                                                           // A(0) -> buffer 0
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_a(j);
    end_a();                                                           // A(1) -> buffer 1
#pragma unroll
    for (int jb = 0; jb < 3; ++jb) issue_w(jb);
    end_w();                                                           // B(0) -> buffer 0
    TD_WAIT_VM_PIECES(0);
    TD_BARRIER_RAW();
    {
        f32x4 x[2][2];
        load_a(0, x);
#pragma unroll
        for (int i = 0; i < 2; ++i) td_split3(x[i][0], x[i][1], ah[i], am[i], al[i]);
        TD_BARRIER_RAW();                                              // every wave has read A(0): its buffer may be refilled (A(2), step 0)
    }
    int cb = 0;                                                        // buffer of B(g); A(g + 1) sits in buffer cb ^ 1
    for (int t = 0; t < my_tiles; ++t) {
        for (int st = 0; st < nsteps; ++st) {
            // step g: twelve fenced groups of four MFMAs; the split of A(g + 1), the fragment reads and this wave's seven DMA pieces
            // (A(g + 2) into the buffer A(g) left, B(g + 1) into the buffer B(g - 1) left) ride between them
            f32x4 nx[2][2];
            load_a(cb ^ 1, nx);
            u32x4 bh0[2], bm0[2], bl0[2], bh1[2], bm1[2], bl1[2], nh[2], nm[2], nl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { bh0[j] = load_b(cb, 0, 0, j); bm0[j] = load_b(cb, 1, 0, j); }
            auto mm = [&](const u32x4 (&A)[2], const u32x4 (&B)[2], int g2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[g2][i][j] = td_mfma32_bf16(A[i], B[j], acc[g2][i][j]);
            };
            auto chunk = [&](int c) {
                const int i = c >> 2, qq = c & 3;
                const float a = qq < 2 ? nx[i][0][2 * qq] : nx[i][1][2 * qq - 4], b = qq < 2 ? nx[i][0][2 * qq + 1] : nx[i][1][2 * qq - 3];
                unsigned h_, m_, l_;
                td_split3_pair(a, b, h_, m_, l_);
                nh[i][qq] = h_; nm[i][qq] = m_; nl[i][qq] = l_;
            };
            TD_SCHED_FENCE();
            mm(ah, bh0, 0); issue_w(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) bl0[j] = load_b(cb, 2, 0, j);
            TD_SCHED_FENCE();
            mm(am, bh0, 0); chunk(0); issue_w(1);
            TD_SCHED_FENCE();
            mm(al, bh0, 0); chunk(1); issue_w(2);
#pragma unroll
            for (int j = 0; j < 2; ++j) bh1[j] = load_b(cb, 0, 1, j);
            TD_SCHED_FENCE();
            mm(ah, bm0, 0); chunk(2); issue_a(0);                      // A(g + 2) goes to the buffer A(g) left (read during step g - 1)
            TD_SCHED_FENCE();
            mm(am, bm0, 0); chunk(3); issue_a(1);
#pragma unroll
            for (int j = 0; j < 2; ++j) bm1[j] = load_b(cb, 1, 1, j);
            TD_SCHED_FENCE();
            mm(ah, bl0, 0); chunk(4); issue_a(2);
            TD_SCHED_FENCE();
            mm(ah, bh1, 1); chunk(5); issue_a(3);
#pragma unroll
            for (int j = 0; j < 2; ++j) bl1[j] = load_b(cb, 2, 1, j);
            TD_SCHED_FENCE();
            mm(am, bh1, 1); chunk(6);
            TD_SCHED_FENCE();
            mm(al, bh1, 1); chunk(7);
            TD_SCHED_FENCE();
            mm(ah, bm1, 1);
            TD_SCHED_FENCE();
            mm(am, bm1, 1);
            TD_SCHED_FENCE();
            mm(ah, bl1, 1);
            TD_SCHED_FENCE();
            end_w(); end_a();
#pragma unroll
            for (int i = 0; i < 2; ++i) { ah[i] = nh[i]; am[i] = nm[i]; al[i] = nl[i]; }
            TD_WAIT_VM_PIECES(0);
            TD_BARRIER_RAW();
            cb ^= 1;
        }
        float* outb = p.out + (size_t)spos.b * p.MP * p.N;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
            td_store_acc<2, 2, ROLE != 0, ROLE == 1>(acc[g2], outb, p.bias, p.resid, p.M, p.N, ROLE == 1 ? 0 : p.act, spos.tm * G::BM + wave * 64,
                                                     spos.tn * G::BN + g2 * 64, lane);
        zero_acc();
        advance(spos);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------------
// fp32 -> bf16 bits, round to nearest even: the arithmetic of v_cvt_pk_bf16_f32 (td_split3 on the device)
static inline unsigned short gemm_b3_bf16(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float gemm_b3_widen(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
static inline int gemm_b3_npad(int N) { return (N + 127) / 128 * 128; }
static inline size_t gemm_b3_packed_bytes(int K, int N) { return (size_t)(K / 16) * 6 * gemm_b3_npad(N) * 16; }
static inline bool gemm_b3_supports(int K, int N) { return K % 16 == 0 && K >= 32 && N % 4 == 0; }
// W[n][k] (one [Cout][Cin] matrix, the layout of a 1x1 conv's OIHW weights / of a Winograd-domain matrix) -> [K/16][part][k-half][NPad][8 bf16];
// packed column `slot` holds output channel  tn*128 + g2*64 + j*2 + nt  for  slot = tn*128 + g2*64 + nt*32 + j  (conv_pack_weights, BN = 128)
static inline void gemm_b3_pack(const float* w, int N, int K, unsigned short* dst) {
    const int NPad = gemm_b3_npad(N);
    for (int st = 0; st < K / 16; ++st)
        for (int slot = 0; slot < NPad; ++slot) {
            const int tn = slot / 128, within = slot % 128, g2 = within / 64, w2 = within % 64, nt = w2 / 32, j = w2 % 32;
            const int n = tn * 128 + g2 * 64 + j * 2 + nt;
            for (int kh = 0; kh < 2; ++kh)
                for (int e = 0; e < 8; ++e) {
                    const float x = n < N ? w[(size_t)n * K + st * 16 + kh * 8 + e] : 0.f;
                    const unsigned short h = gemm_b3_bf16(x);
                    const float r = x - gemm_b3_widen(h);
                    const unsigned short m = gemm_b3_bf16(r);
                    const unsigned short l = gemm_b3_bf16(r - gemm_b3_widen(m));
                    const unsigned short parts[3] = {h, m, l};
                    for (int part = 0; part < 3; ++part)
                        dst[((((size_t)st * 3 + part) * 2 + kh) * NPad + slot) * 8 + e] = parts[part];
                }
        }
}

// wr = 4: 256 x 128 tiles, one workgroup per CU; wr = 2: 128 x 128 tiles, two per CU.  grid_cap > 0 forces the number of workgroups (tests).
template <int VAR>
static inline void gemm_b3_launch_v(const GemmArgs& a, int wr, int role, long grid, hipStream_t s) {
    if (wr == 4) {
        if (role == 1) TD_LAUNCH((k_gemm_b3<4, 1, VAR>), dim3((unsigned)grid), dim3(512), GemmB3Geom<4>::LDS_BYTES, s, a);
        else if (role == 2) TD_LAUNCH((k_gemm_b3<4, 2, VAR>), dim3((unsigned)grid), dim3(512), GemmB3Geom<4>::LDS_BYTES, s, a);
        else TD_LAUNCH((k_gemm_b3<4, 0, VAR>), dim3((unsigned)grid), dim3(512), GemmB3Geom<4>::LDS_BYTES, s, a);
    } else {
        if (role == 1) TD_LAUNCH((k_gemm_b3<2, 1, VAR>), dim3((unsigned)grid), dim3(256), GemmB3Geom<2>::LDS_BYTES, s, a);
        else if (role == 2) TD_LAUNCH((k_gemm_b3<2, 2, VAR>), dim3((unsigned)grid), dim3(256), GemmB3Geom<2>::LDS_BYTES, s, a);
        else TD_LAUNCH((k_gemm_b3<2, 0, VAR>), dim3((unsigned)grid), dim3(256), GemmB3Geom<2>::LDS_BYTES, s, a);
    }
}
static inline void gemm_b3_launch(GemmArgs a, int wr, int grid_cap, hipStream_t s) {
    if (wr == 1) {                                                     // 256 x 128 tiles, no loader waves, two workgroups per CU (k_gemm_b3m)
        a.NPad = gemm_b3_npad(a.N);
        a.tiles_m = (a.M + 255) / 256;
        a.tiles_n = a.NPad / 128;
        const long total = (long)a.tiles_m * a.tiles_n * a.nbatch;
        long grid = grid_cap > 0 ? grid_cap : 512;
        if (grid > total) grid = total;
        const int role = (a.nbatch > 1 && !a.wshare) ? 1 : a.resid ? 0 : 2;
        if (role == 1) TD_LAUNCH((k_gemm_b3m<1>), dim3((unsigned)grid), dim3(256), GemmB3mGeom::LDS_BYTES, s, a);
        else if (role == 2) TD_LAUNCH((k_gemm_b3m<2>), dim3((unsigned)grid), dim3(256), GemmB3mGeom::LDS_BYTES, s, a);
        else TD_LAUNCH((k_gemm_b3m<0>), dim3((unsigned)grid), dim3(256), GemmB3mGeom::LDS_BYTES, s, a);
        return;
    }
    const int BM = 64 * wr;
    a.NPad = gemm_b3_npad(a.N);
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = a.NPad / 128;
    const long total = (long)a.tiles_m * a.tiles_n * a.nbatch;
    long grid = grid_cap > 0 ? grid_cap : (wr == 4 ? 256 : 512);
    if (grid > total) grid = total;
    const int role = (a.nbatch > 1 && !a.wshare) ? 1 : a.resid ? 0 : 2;
#ifdef TD_B3_PROBE                                                     // tools/b3_probe.py: schedule variant / skip flags of the 256-row Winograd GEMM from the environment
    if (wr == 4 && role == 1) {
        const char* ev = getenv("TD_B3_VAR");
        const char* es = getenv("TD_B3_SKIP");
        const int var = ev ? atoi(ev) : GEMM_B3_VAR, skip = es ? atoi(es) : 0;
#define TD_B3_CASE(V, S) if (var == V && skip == S) { TD_LAUNCH((k_gemm_b3<4, 1, V, S>), dim3((unsigned)grid), dim3(512), GemmB3Geom<4>::LDS_BYTES, s, a); return; }
        TD_B3_CASE(0, 0) TD_B3_CASE(1, 0) TD_B3_CASE(8, 0) TD_B3_CASE(0, 64) TD_B3_CASE(0, 66) TD_B3_CASE(0, 67) TD_B3_CASE(0, 125) TD_B3_CASE(0, 3) TD_B3_CASE(0, 61)
#undef TD_B3_CASE
        fprintf(stderr, "td_gemm_b3: variant %d / skip %d is not instantiated\n", var, skip);
    }
#endif
    gemm_b3_launch_v<GEMM_B3_VAR>(a, wr, role, grid, s);
}
// test / probe hook: a forced ConvTile code chooses the kernel form (64-row codes: 128-row tiles with loaders, 64-column codes: no loader waves)
static inline int gemm_b3_wr_of_tile(int tile) { const ConvTileDims d = conv_tile_dims((ConvTile)tile); return d.BM == 64 ? 2 : d.BN == 64 ? 1 : 4; }
// Which kernel form for a GEMM of `rows` x N per batch: 0 = none -- fewer than one 256 x 128 tile per CU, where the exact-fp32 kernels with
// their 64 / 128-row tiles are faster (512 -> 512 on 2048 rows: 22 us against 37; 512 -> 64: 24 against 40; profiles/r06a_*) --, else 1 = the
// matrix-only form (k_gemm_b3m; 0.308 ms against 0.322 for the loader-wave form on layer 4's 512 -> 512 conv, profiles/r06e_*).
static inline int gemm_b3_pick_wr(long rows, int nbatch, int N) {
    const long tiles = ((rows + 255) / 256) * (gemm_b3_npad(N) / 128) * nbatch;
    return tiles >= 256 && N >= 128 ? 1 : 0;
}
