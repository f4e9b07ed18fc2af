// td_gemm_dma.h -- the batched fp32-MFMA GEMM of the Winograd convs (td_gemm.h ROLE 1) with both operands fed by LDS-DMA, and
// optionally Winograd transforms of OTHER data riding in the instruction stream of its matrix waves.
//
// Why.  The transforms of a Winograd conv are HBM-bound, its GEMMs MFMA-bound, and on one stream they run in series:
// 0.6 ms of a 3.7 ms frame with idle matrix pipes.  The workgroup dispatcher does not overlap them (a kernel that arrives while the
// persistent GEMM holds the CUs is not admitted beside it), so the overlap has to be built INTO the GEMM's workgroups.  First attempt: a
// fifth "rider" wave per workgroup doing transform units between the barriers (needs <= 96 VGPRs for three 5-wave workgroups per CU,
// tools/occupancy_probe.hip; fed by DMA the GEMM needs 82).  Structure free (idle riders cost nothing), but a wave on a SIMD whose
// matrix pipe is saturated gets ~1 VALU instruction per 26 cycles: 4.9 us per unit against a K step of 2.9 us, the rider paced the
// barriers and every launch took the transform's stand-alone time longer (profiles/r03n_*).  A wave's OWN MFMA, though, leaves it ~13
// issue slots per 64-cycle MFMA on paper, so the second attempt (this file, TT = 1 / 2) lets each MATRIX wave carry one transform unit at
// a time: loads requested at the top of a K step, arithmetic and stores in the next step.  MEASURED (profiles/r03n_*): the same +22 us
// per launch (GA 116, GB 71 us against 94 / 49 with the units disabled), and spreading the units evenly over the launch made it worse
// (134 / 83: every step of every workgroup then has one late wave).  The cost is the transform's ~350 instructions per unit at ~22
// cycles each wherever they run: the fp32 MFMA occupies its SIMD's issue port for its whole 64 cycles, and at 87 % matrix-pipe
// utilisation a fifth of the cycles is all the VALU gets.  One channel per lane makes the transforms VALU-heavy (33 VALU per KB; the
// stand-alone 4-channel kernels need 8 and are HBM-bound), and the 4-channel form does not fit the registers of a GEMM wave.  So on
// this chip the transforms of an fp32 Winograd conv cannot hide under its own kind of GEMM: TT = 1 / 2 stay as a measured, opt-in
// experiment (tdnet_opts.overlap bit 64); the default uses TT = 0, which is simply the faster GEMM (82 VGPRs, +2.6 % over td_gemm.h).
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
#include "td_wino.h"

struct GemmDmaGeom {
    static constexpr int BM = 64, BN = 128;
    static constexpr int A_BYTES = BM * 128, B_BYTES = 8 * BN * 16, BUF_BYTES = A_BYTES + B_BYTES, LDS_BYTES = 2 * BUF_BYTES;
};

// What rides in a launch: units [u0, u1) of an input transform OR of an output transform (td_wino.h WinoArgs of the chunk they belong
// to; a unit = one (tile, 64-channel slice), one channel per lane).  Wave v of workgroup w takes the units 4 w + v, + 4 G, + 8 G, ...
struct RiderArgs {
    WinoArgs tin, tout;
    int in_u0, in_u1, out_u0, out_u1;
};

// TT: 0 = GEMM only; 1 = units [in_u0, in_u1) of the input transform rw.tin ride along; 2 = units [out_u0, out_u1) of rw.tout
template <int TT>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 3) k_gemm_dma(GemmArgs p, RiderArgs rw) {
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

    // ---- the riding transform: this wave's units u_next, u_next + u_stride, ... ------------------------------------------------
    WinoInRide ride_i;
    WinoOutRide ride_o;
    bool ride_pending = false;                                        // loads of a unit requested, not yet finished
    const int u_stride = (int)gridDim.x * 4;
    int u_next = (TT == 1 ? rw.in_u0 : rw.out_u0) + (int)blockIdx.x * 4 + wave;
    const int u_end = TT == 1 ? rw.in_u1 : TT == 2 ? rw.out_u1 : 0;
    auto ride_issue = [&]() {
        if (TT == 0 || u_next >= u_end) return;                       // wave-uniform
        if (TT == 1) td_wino4_in_issue(rw.tin, u_next, ride_i); else td_wino4_out_issue(rw.tout, u_next, ride_o);
        u_next += u_stride;
        ride_pending = true;
    };
    auto ride_finish = [&]() {
        if (TT == 1) td_wino4_in_finish(rw.tin, ride_i); else if (TT == 2) td_wino4_out_finish(rw.tout, ride_o);
        ride_pending = false;
    };
    if (my_tiles == 0) {                                              // no tile for this workgroup: its waves still do their units
        while (TT && u_next < u_end) { ride_issue(); ride_finish(); }
        return;
    }

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
    auto compute = [&](int buf, int ibuf, auto with_ride) {
        constexpr bool RIDE = TT != 0 && decltype(with_ride)::value;
        const char* base = smem + buf * G::BUF_BYTES;
        f32x4 af[2], bf[2][2];
        // The riding unit's arithmetic and stores: independent of everything below.  Left to itself the compiler puts all ~350 of those
        // instructions AHEAD of the step's first MFMA (the wave then reaches the barrier 2-3 k cycles late and paces its workgroup); the
        // scheduling groups at the end of this block pin them BETWEEN the MFMAs, ten VALU and a store per MFMA.
        if constexpr (RIDE) ride_finish();
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
                if constexpr (!RIDE) TD_SCHED_FENCE();
                if (2 * g + h2 < 6) issue_piece(ibuf, 2 * g + h2);
                if constexpr (!RIDE) TD_SCHED_FENCE();
            }
        }
        if constexpr (RIDE) {
            TD_SCHED_GROUP(0x100, 3);                                 // the first fragments
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        TD_SCHED_GROUP(0x008, 1);                     // one MFMA ...
                        TD_SCHED_GROUP(0x002, 10);                    // ... ten VALU of the riding unit in its shadow ...
                        TD_SCHED_GROUP(0x040, 1);                     // ... and one of its stores
                    }
                    TD_SCHED_GROUP(0x020, 1);                         // the DMA piece of this half group
                    if (h2 == 0 && g < 3) TD_SCHED_GROUP(0x100, 3);   // the next group's fragments
                }
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
            if (TT != 0 && ride_pending) compute(cb, cb ^ 1, std::true_type{});      // wave-uniform
            else compute(cb, cb ^ 1, std::false_type{});
            TD_WAIT_VM_PIECES(0);
            TD_BARRIER_RAW();
            cb ^= 1;
            ride_issue();                                             // the next unit's loads: a whole K step ahead of their use
        }
        float* outb = p.out + (size_t)spos.b * p.MP * p.N;
        td_store_acc<1, 2, true, true>(acc, outb, p.bias, nullptr, p.M, p.N, 0, spos.tm * G::BM + wm * 32, spos.tn * G::BN + wn * 64, lane);
        zero_acc();
        advance(spos);
    }
    if (TT != 0) {                                                    // units left over when the tiles ran out (short launches): unhidden tail
        if (ride_pending) ride_finish();
        while (u_next < u_end) { ride_issue(); ride_finish(); }
    }
}

static inline bool gemm_dma_supports(int K, int N, ConvTile tile) {
    const ConvTileDims d = conv_tile_dims(tile);                      // weights packed for a BN = 128 / two-wave-column tile
    return K % 32 == 0 && d.BN == 128 && d.WGN == 2 && N % 4 == 0;
}
// grid_cap > 0 forces the number of workgroups (tests); rw == nullptr: four waves, no rider
// rw: one transform rides (its input OR its output units; a launch carries one kind); nullptr = GEMM only
static inline void gemm_dma_launch(GemmArgs a, const RiderArgs* rw, int grid_cap, hipStream_t s) {
    a.tiles_m = (a.M + 63) / 64;
    a.tiles_n = a.NPad / 128;
    const long total = (long)a.tiles_m * a.tiles_n * a.nbatch;
    long grid = grid_cap > 0 ? grid_cap : 768;
    if (grid > total) grid = total;
    // (A grid with the same number of tiles for every workgroup -- 576 instead of 768 for 1152 tiles -- measured 1 % SLOWER in the frame,
    // with and without the chains: 269.2 -> 266.5, 267.0 -> 264.9.  The half-empty last round overlaps the next launch's ramp.)
    if (rw && rw->in_u1 > rw->in_u0) TD_LAUNCH((k_gemm_dma<1>), dim3((unsigned)grid), dim3(256), GemmDmaGeom::LDS_BYTES, s, a, *rw);
    else if (rw && rw->out_u1 > rw->out_u0) TD_LAUNCH((k_gemm_dma<2>), dim3((unsigned)grid), dim3(256), GemmDmaGeom::LDS_BYTES, s, a, *rw);
    else {
        RiderArgs none;
        none.in_u0 = none.in_u1 = none.out_u0 = none.out_u1 = 0;
        TD_LAUNCH((k_gemm_dma<0>), dim3((unsigned)grid), dim3(256), GemmDmaGeom::LDS_BYTES, s, a, none);
    }
}
