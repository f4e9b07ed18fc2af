// td_gemm_dma.h -- the batched fp32-MFMA GEMM of the Winograd convs (td_gemm.h ROLE 1) with both operands fed by LDS-DMA: no staging
// registers (82 VGPRs, three workgroups per CU), +2.6 % in the frame over td_gemm.h (profiles/r03m_*).
//
// (Rounds 3-4 also carried two forms of this kernel in which Winograd transforms of OTHER data rode in a fifth wave / in the matrix
// waves' own instruction stream -- TT = 1 / 2, tdnet_opts.overlap bit 64.  Measured -4 % in the frame: the fp32 MFMA occupies its SIMD's
// issue port for its whole 64 cycles and leaves the VALU a fifth of them (DESIGN_experiments 4.1d, profiles/r03n_*).  Removed in round 5;
// last commit with that code: 78dfa5a.)
//
//   out[b][m][n] = sum_k A[b][m][k] W[b][k][n]      (plain epilogue: bias, residual and activation belong to the output transform)
//
// Tile 64 x 128, K step 32, four matrix waves as 2 x 2 (32 x 64 each), two LDS buffers of 24 KB (three workgroups per CU):
//   A: [row][32 floats] = full 128-byte lines of the row-major operand (8 lanes per row, 8 rows per DMA piece), 16-byte slots XOR-
//      swizzled with (row >> 1) & 7 on the SOURCE address, so that the fragment reads (lanes = consecutive rows) are conflict-free;
//   B: the packed weights [kq][n][4 floats] of td_conv.h, a linear copy.
// Same k-permutation inside a step as k_conv_igemm / k_gemm_persistent (slot kq = floats 4 kq .. 4 kq + 3; lane half h takes slot
// 2 g + h in MFMA group g), same output-column permutation, same accumulation order: results are bit-identical to td_gemm.h.
// The DMA pieces of step s + 1 are issued between the MFMA groups of step s (an LDS-DMA instruction costs its wave 100-150 cycles of
// issue: td_conv_hd.h), across tile boundaries (persistent tile list as in td_gemm.h); one bare barrier per step.
#pragma once
#include "td_gemm.h"

struct GemmDmaGeom {
    static constexpr int BM = 64, BN = 128;
    static constexpr int A_BYTES = BM * 128, B_BYTES = 8 * BN * 16, BUF_BYTES = A_BYTES + B_BYTES, LDS_BYTES = 2 * BUF_BYTES;
};

TD_KERNEL void TD_LAUNCH_BOUNDS(256, 3) k_gemm_dma(GemmArgs p) {
    using G = GemmDmaGeom;
    TD_DYN_LDS(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int nsteps = p.K >> 5;

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

    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
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
    const unsigned w_step_bytes = 8u * (unsigned)p.NPad * 16u;
    const unsigned a_bytes = (unsigned)p.M * (unsigned)p.K * 4u, w_bytes = (unsigned)nsteps * w_step_bytes;

    // ---- loader state: the (tile, step) whose pieces are issued next.  A piece pa = wave + 4 j: rows 8 pa .. + 7, lane -> row + (l >> 3),
    // LDS slot l & 7 = floats 4 kq .. of the step's 32 with kq = slot ^ ((row >> 1) & 7).  B piece pb = wave + 4 jb = 2 kq + nh. --------
    int l_tile = 0, l_step = 0;
    TilePos lpos = pos0;
    TdBuf a_buf, w_buf;
    unsigned a_off[2], b_off[4];
    auto loader_enter_tile = [&]() {
        a_buf = td_make_buf(p.a + (size_t)lpos.b * p.MP * p.K, a_bytes);
        w_buf = td_make_buf(p.wp + (size_t)lpos.b * nsteps * 8 * p.NPad * 4, w_bytes);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * (wave + 4 * j) + (lane >> 3);
            const int m = lpos.tm * G::BM + row;
            const int kq = (lane & 7) ^ ((row >> 1) & 7);
            a_off[j] = m < p.M ? ((unsigned)m * (unsigned)p.K + (unsigned)kq * 4u) * 4u : TD_BUF_OOB;
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int pb = wave + 4 * jb;
            b_off[jb] = (unsigned)((pb >> 1) * p.NPad + lpos.tn * G::BN + (pb & 1) * 64 + lane) * 16u;
        }
    };
    auto issue_piece = [&](int buf, int pc) {                         // pc = 0..5, compile time
        char* base = smem + buf * G::BUF_BYTES;
        if (pc < 2) td_buf_ld16_lds(a_buf, base + (wave + 4 * pc) * 1024, a_off[pc], (unsigned)l_step * 128u);
        else td_buf_ld16_lds(w_buf, base + G::A_BYTES + (wave + 4 * (pc - 2)) * 1024, b_off[pc - 2], (unsigned)l_step * w_step_bytes);
    };
    auto issue_end = [&]() {
        if (++l_step == nsteps) {
            l_step = 0;
            if (++l_tile < my_tiles) { advance(lpos); loader_enter_tile(); }   // past the end: stay on the last tile (never consumed)
        }
    };

    // ---- fragment addresses (bytes inside a buffer) ------------------------------------------------------------------------------
    unsigned a_rd[4];
    {
        const int row = wm * 32 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) a_rd[g] = (unsigned)(row * 128 + (((2 * g + half) ^ ((row >> 1) & 7)) << 4));
    }
    const unsigned b_rd = (unsigned)(G::A_BYTES + half * 2048 + (wn * 64 + l31) * 16);

    f32x16 acc[1][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    };
    // one K step on buffer `buf`; the six pieces of the next step go out after the MFMA half-groups (eight slots of four MFMAs)
    auto compute = [&](int buf, int ibuf) {
        const char* base = smem + buf * G::BUF_BYTES;
        f32x4 af[2], bf[2][2];
        af[0] = *reinterpret_cast<const f32x4*>(base + a_rd[0]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(base + b_rd + j * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
                af[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(base + a_rd[g + 1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f32x4*>(base + b_rd + (g + 1) * 4096 + j * 512);
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                for (int s = 2 * h2; s < 2 * h2 + 2; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[0][j] = td_mfma32(af[g & 1][s], bf[g & 1][j][s], acc[0][j]);
                TD_SCHED_FENCE();
                if (2 * g + h2 < 6) issue_piece(ibuf, 2 * g + h2);
                TD_SCHED_FENCE();
            }
        }
        issue_end();
    };

    TilePos spos = pos0;                                              // the tile being multiplied
    zero_acc();
    loader_enter_tile();
#pragma unroll
    for (int pc = 0; pc < 6; ++pc) issue_piece(0, pc);
    issue_end();
    TD_WAIT_VM_PIECES(0);
    TD_BARRIER_RAW();
    int cb = 0;
    for (int t = 0; t < my_tiles; ++t) {
        for (int st = 0; st < nsteps; ++st) {
            compute(cb, cb ^ 1);
            TD_WAIT_VM_PIECES(0);
            TD_BARRIER_RAW();
            cb ^= 1;
        }
        float* outb = p.out + (size_t)spos.b * p.MP * p.N;
        td_store_acc<1, 2, true, true>(acc, outb, p.bias, nullptr, p.M, p.N, 0, spos.tm * G::BM + wm * 32, spos.tn * G::BN + wn * 64, lane);
        zero_acc();
        advance(spos);
    }
}

static inline bool gemm_dma_supports(int K, int N, ConvTile tile) {
    const ConvTileDims d = conv_tile_dims(tile);                      // weights packed for a BN = 128 / two-wave-column tile
    return K % 32 == 0 && d.BN == 128 && d.WGN == 2 && N % 4 == 0;
}
// grid_cap > 0 forces the number of workgroups (tests)
static inline void gemm_dma_launch(GemmArgs a, int grid_cap, hipStream_t s) {
    a.tiles_m = (a.M + 63) / 64;
    a.tiles_n = a.NPad / 128;
    const long total = (long)a.tiles_m * a.tiles_n * a.nbatch;
    long grid = grid_cap > 0 ? grid_cap : 768;
    if (grid > total) grid = total;
    // (A grid with the same number of tiles for every workgroup -- 576 instead of 768 for 1152 tiles -- measured 1 % SLOWER in the frame,
    // with and without the chains: 269.2 -> 266.5, 267.0 -> 264.9.  The half-empty last round overlaps the next launch's ramp.)
    TD_LAUNCH(k_gemm_dma, dim3((unsigned)grid), dim3(256), GemmDmaGeom::LDS_BYTES, s, a);
}
