// td_conv_hd.h -- the fp16-MFMA implicit-GEMM convolution fed by LDS-DMA (`buffer_load_dwordx4 ... lds`), for the convs whose input
// map is already fp16 in HBM (tdnet_opts.precision = 1: every conv inside the backbone; BASELINE.json config 5).
//
// Why a second kernel.  k_conv_igemm_h (td_conv_h.h) stages a 128 x 128 x 64 step through registers: 32 KB per 64 MFMAs.  On the fp16
// pipe an MFMA is 32 cycles per SIMD, so a CU retires that step in 512 cycles -- and the vector-memory path delivers 64 B/clk per
// CU: 512 cycles for the same 32 KB.  Loads and MFMAs each need the whole step, and the register -> LDS write path (ds_write_b128,
// ~79 B/clk per CU) needs another 80 % of it: the kernel tops out at 0.37 of the fp16 roof however it is scheduled.  Here:
//   * block tile 64 RH x 128 (RH = 4, 3, 2: 256 / 192 / 128 rows), 2 RH waves of 64 x 64: at RH = 4 a step moves 48 KB for 128 MFMAs
//     (0.75 of the memory path per MFMA-bound step instead of 1.0), one workgroup of 8 waves per CU;
//   * both operands go global -> LDS by DMA: no staging registers, no ds_write, the loads of step s + 2 are issued before the MFMAs
//     of step s into a ring of 3 LDS buffers (NBUF = 3) and waited for with a COUNTED vmcnt, so two steps of memory latency are
//     covered; the barrier is the bare s_barrier (a __syncthreads() would drain the DMA queue);
//   * the activation operand arrives in FULL 128-byte lines (8 lanes = the 64 channels of one pixel, 8 pixels per wave instruction;
//     lane-per-pixel pieces touch 64 lines for 1 KB and run the address path 4-8x longer) into a row-major image [row][64 halfs]
//     whose 16-byte slots are XOR-swizzled with (row >> 1) & 7 -- applied to the SOURCE address, the DMA writes lane-linear -- so the
//     MFMA fragment reads (one ds_read_b128 per operand per MFMA, lanes = consecutive rows) are bank-conflict free;
//   * RH = 3 exists for the tile count: at 720 x 960 (90 x 120 = 10800 pixels) 192-row tiles make 57 x 4 = 228 workgroups for 256
//     CUs, where 128 x 128 tiles were 340 on 512 slots (a round and a third).
// Padding taps and rows past M are out-of-range buffer offsets: the DMA writes zeros for them, like a register load returns zeros.
// Weights: the [step][kq][CoutPad][8 halfs] packing of td_conv_h.h for a BN = 128 / two-wave-column tile, unchanged.
#pragma once
#include "td_conv_h.h"

template <int RH, int NB = 1>
struct ConvDmaGeom {
    static constexpr int BM = 64 * RH, BN = 128 * NB, NW = 2 * RH;
    static constexpr int A_BYTES = BM * 128, B_BYTES = 8 * BN * 16, BUF_BYTES = A_BYTES + B_BYTES;
    static constexpr int NPA = BM / 8, NPB = 16 * NB;               // 1 KB DMA pieces per step: A = 8 pixels x 128 B, B = 64 packed weight slots
    static constexpr int NPW = (NPA + NPB + NW - 1) / NW;           // pieces per wave and step (RH = 3: 42 for 40, two waves repeat a piece)
    static_assert(NPA % NW == 0 && NPA / NW == 4, "every wave stages four A pieces per step");
};

// NB = 2: 256 output channels per tile (a wave multiplies 64 rows x 128 channels = two 64-slot groups of the packed weights): 64 KB
// per K step for 256 MFMAs -- 31 bytes per clock and CU from L2 at the full MFMA rate, against 47 for the 256 x 128 tile and 62
// for 128 x 128: with every CU streaming, the L2 -> CU fabric is what these kernels run into first.
template <int RH, int KS, bool OUT16, int NBUF, int NB = 1>
TD_KERNEL void TD_LAUNCH_BOUNDS(128 * RH, 1) k_conv_dma_h(ConvArgs p) {
    using G = ConvDmaGeom<RH, NB>;
    static_assert(NB == 1 || (NB == 2 && RH == 4 && NBUF == 2), "256-channel tiles: 256 rows, two LDS buffers");
    constexpr int NJ = 2 * NB;                                      // 32-column accumulators per wave
    constexpr int BM = G::BM, NW = G::NW, NPW = G::NPW, NTAPS = KS * KS;
    static_assert(NBUF == 2 || NBUF == 3, "ring of 2 or 3 LDS buffers");
    TD_DYN_LDS(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * G::BN;

    // ---- DMA geometry: A piece j of this wave = rows 8 (wave + NW j) .. + 7; lane l -> row + (l >> 3), LDS slot l & 7, which holds
    // the channels 8 kq .. 8 kq + 7 of the chunk with kq = slot ^ ((row >> 1) & 7) ------------------------------------------------
    unsigned a_off[4], a_taps[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wave + NW * j) + (lane >> 3);
        const int m = m0 + row;
        const int oy = m / p.Wo, ox = m - oy * p.Wo;
        const int by = (m < p.M) ? oy * p.stride - p.pad : -(1 << 28), bx = ox * p.stride - p.pad;
        const int kq = (lane & 7) ^ ((row >> 1) & 7);
        a_off[j] = (((unsigned)by * (unsigned)p.W + (unsigned)bx) * (unsigned)p.Cin + (unsigned)kq * 8u) * 2u;
        a_taps[j] = 0u;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int iy = by + (t / KS) * p.dil, ix = bx + (t % KS) * p.dil;
            a_taps[j] |= ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (1u << t) : 0u;
        }
    }
    const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 2u);
    const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
    const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;
    // B piece pb = (BN / 64) kq + q (64 consecutive packed slots of k-group kq); this wave stages pb = wave + NW jb
    unsigned b_off[NPW - 4];
#pragma unroll
    for (int jb = 0; jb < NPW - 4; ++jb) {
        int pb = wave + NW * jb;
        if (pb >= G::NPB) pb -= NW;                                  // RH = 3: the two surplus slots repeat a piece (same bytes, same place)
        b_off[jb] = (unsigned)((pb / (2 * NB)) * p.CoutPad + n0 + (pb % (2 * NB)) * 64 + lane) * 16u;
    }

    int l_step = 0, l_chunk = 0, l_tap = 0;
    auto issue = [&](int buf) {                                       // all NPW pieces of the next K step -> LDS buffer `buf`
        char* base = smem + buf * G::BUF_BYTES;
        const int ky = l_tap / KS;
        const int dy = ky * p.dil, dx = (l_tap - ky * KS) * p.dil;
        const unsigned delta = (unsigned)((dy * p.W + dx) * p.Cin + l_chunk * 64) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = ((a_taps[j] >> l_tap) & 1u) != 0u;
            td_buf_ld16_lds(in_buf, base + (wave + NW * j) * 1024, ok ? a_off[j] + delta : TD_BUF_OOB, 0u);
        }
        const unsigned wsoff = (unsigned)(l_step < p.nsteps ? l_step : p.nsteps - 1) * w_step_bytes;
#pragma unroll
        for (int jb = 0; jb < NPW - 4; ++jb) {
            int pb = wave + NW * jb;
            if (pb >= G::NPB) pb -= NW;
            td_buf_ld16_lds(w_buf, base + G::A_BYTES + pb * 1024, b_off[jb], wsoff);
        }
        ++l_step;
        if (++l_tap == NTAPS) { l_tap = 0; ++l_chunk; }
    };

    // ---- MFMA fragment addresses (bytes inside a buffer) -----------------------------------------------------------------
    unsigned a_rd[2][4];                                              // [i][g]: row wm 64 + 32 i + l31, k-group 2 g + half, swizzled slot
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + 32 * i + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) a_rd[i][g] = (unsigned)(row * 128 + (((2 * g + half) ^ ((row >> 1) & 7)) << 4));
    }
    constexpr int BKQ = G::BN * 16;                                   // bytes per k-group of the weight image
    const unsigned b_rd = (unsigned)(G::A_BYTES + half * BKQ + (wn * 64 * NB + l31) * 16);

    f32x16 acc[2][NJ];                                                // [i][2 sb + nt]: rows 32 i .., packed slots wn 64 NB + 64 sb + 32 nt ..
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const char* base = smem + buf * G::BUF_BYTES;
        f16x8 af[2][2], bf[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(base + a_rd[i][0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(base + b_rd + j * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const f16x8*>(base + a_rd[i][g + 1]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f16x8*>(base + b_rd + (g + 1) * 2 * BKQ + j * 512);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = td_mfma32_f16(af[g & 1][i], bf[g & 1][j], acc[i][j]);
        }
    };

    // ---- ring: the DMA of step s + NBUF - 1 is issued before the MFMAs of step s; a counted wait leaves the newest step(s) in
    // flight.  Order per step: issue, compute, wait for the NEXT step's pieces (this wave's), barrier (everyone's have landed, and
    // everyone is done reading the buffer the next issue overwrites).
    issue(0);
    if (NBUF == 3) issue(1);
    if (NBUF == 3) TD_WAIT_VM_PIECES(NPW); else TD_WAIT_VM_PIECES(0);
    TD_BARRIER_RAW();
    int cb = 0, ib = NBUF - 1;                                        // buffer being multiplied / buffer being filled
    for (int step = 0; step < p.nsteps; ++step) {
        issue(ib);                                                    // past the last step: a harmless surplus tile (clamped weights, zeros)
        compute(cb);
        if (NBUF == 3) TD_WAIT_VM_PIECES(NPW); else TD_WAIT_VM_PIECES(0);
        TD_BARRIER_RAW();
        cb = cb + 1 == NBUF ? 0 : cb + 1;
        ib = ib + 1 == NBUF ? 0 : ib + 1;
    }
    TD_WAIT_VM_PIECES(0);                                             // the surplus pieces must not land in an LDS that has been handed on

    if constexpr (NB == 1) {
        td_store_acc_h<2, 2, OUT16, true>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * 64, n0 + wn * 64, lane);
    } else {
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) {                            // each 64-slot group is one wave-column of the weight packing: its own epilogue
            f32x16 part[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) part[i][nt] = acc[i][2 * sb + nt];
            td_store_acc_h<2, 2, OUT16, true>(part, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * 64, n0 + wn * 64 * NB + sb * 64, lane);
        }
    }
}

// rows per tile / 64 for an output of M pixels x Cout channels: the launch ends when the busiest CU is done, one workgroup per CU
// (RH = 4, 3) or two (RH = 2, two LDS buffers); relative tile efficiencies from the memory-path arithmetic in the header comment.
// returns rh, or 8 for the 256 x 256 tile (CoutPad must be a multiple of 256 for it: the caller checks)
static inline int conv_dma_pick_rh(long M, int Cout, bool allow256 = true) {
    int best = 4;
    double best_cost = 0.0;
    static const struct { int rh, nb; double eff; int per_cu; } cand[4] = {{4, 1, 1.00, 1}, {3, 1, 0.94, 1}, {2, 1, 0.80, 2}, {4, 2, 1.25, 1}};
    for (int i = 0; i < 4; ++i) {
        if (cand[i].nb == 2 && (!allow256 || Cout % 256)) continue;
        const long tn = (Cout + 128 * cand[i].nb - 1) / (128 * cand[i].nb);
        const long tiles = ((M + 64 * cand[i].rh - 1) / (64 * cand[i].rh)) * tn, slots = 256L * cand[i].per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * cand[i].per_cu * cand[i].rh * cand[i].nb / cand[i].eff;
        if (i == 0 || cost < best_cost) { best_cost = cost; best = cand[i].nb == 2 ? 8 : cand[i].rh; }
    }
    return best;
}
static inline bool conv_dma_supports(int Cin, int Cout, int KS, ConvTile tile) {
    const ConvTileDims d = conv_tile_dims(tile);
    return Cin % 64 == 0 && Cout >= 128 && d.BN == 128 && d.WGN == 2 && (KS == 1 || KS == 3);
}

template <int RH, int NBUF, int NB>
static inline void conv_launch_dma_t(const ConvArgs& a, int KS, bool out16, hipStream_t s) {
    using G = ConvDmaGeom<RH, NB>;
    const int grid = ((a.M + G::BM - 1) / G::BM) * a.tiles_n;
    const int lds = NBUF * G::BUF_BYTES;
    if (KS == 3 && out16) TD_LAUNCH((k_conv_dma_h<RH, 3, true, NBUF, NB>), dim3(grid), dim3(128 * RH), lds, s, a);
    else if (KS == 3) TD_LAUNCH((k_conv_dma_h<RH, 3, false, NBUF, NB>), dim3(grid), dim3(128 * RH), lds, s, a);
    else if (out16) TD_LAUNCH((k_conv_dma_h<RH, 1, true, NBUF, NB>), dim3(grid), dim3(128 * RH), lds, s, a);
    else TD_LAUNCH((k_conv_dma_h<RH, 1, false, NBUF, NB>), dim3(grid), dim3(128 * RH), lds, s, a);
}
// rh: 4 / 3 (three LDS buffers, one workgroup per CU), 2 (two buffers, two per CU), 8 = 256 rows x 256 channels (two buffers, one per CU)
static inline void conv_launch_dma(ConvArgs a, int rh, int KS, bool out16, hipStream_t s) {
    a.tiles_n = a.CoutPad / (rh == 8 ? 256 : 128);
    if (rh == 8) conv_launch_dma_t<4, 2, 2>(a, KS, out16, s);
    else if (rh == 4) conv_launch_dma_t<4, 3, 1>(a, KS, out16, s);
    else if (rh == 3) conv_launch_dma_t<3, 3, 1>(a, KS, out16, s);
    else conv_launch_dma_t<2, 2, 1>(a, KS, out16, s);
}
