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

// MI = 32-row accumulator blocks per wave: 2 = waves of 64 x 64 (2 RH waves), 1 = waves of 32 x 64 (4 RH waves: two per SIMD at RH = 2,
// for the grids that put a single workgroup on a CU -- a wave's DMA issue is then covered by the other wave of its SIMD).
constexpr int TD_CUS = 256;                                            // MI355X: 8 XCDs x 32 CUs
// Tile / K-loop form of the LDS-DMA conv kernels (ConvLayer::rh).  The model uses NONE, 128, 192, 256 and 256x256 (conv_dma_pick_rh);
// the others force one form for the probes and tests (tile code 16 + ... of tdnet_op_conv2d_f16io, see there).
enum ConvDmaCode {
    CD_NONE = 0,          // not on these kernels
    CD_128 = 2,           // 128 x 128: buffers and waves by the grid (see conv_launch_dma)
    CD_192 = 3,           // 192 x 128
    CD_256 = 4,           // 256 x 128
    CD_128_2BUF = 5,      // 128 x 128, two LDS buffers, four waves (two workgroups per CU)
    CD_128_4BUF = 6,      // 128 x 128, ring of four, four waves of 64 x 64
    CD_128_8W = 7,        // 128 x 128, ring of four, eight waves of 32 x 64
    CD_256x256 = 8,       // 256 x 256 (Cout padded to a multiple of 256)
    CD_128_SUPER = 9,     // row-image kernel, 128 rows: one barrier per super-step
    CD_192_SUPER = 10,    // ... 192 rows
    CD_192_STEP = 11,     // row-image kernel, 192 rows: one barrier per K step, three weight buffers
    CD_128_STEP = 12,     // ... 128 rows
    CD_256_EARLY = 13,    // row-image kernel, 256 rows: four weight buffers, data lands one step early
    CD_192_EARLY = 14,    // ... 192 rows
    CD_128_EARLY = 15,    // ... 128 rows
    CD_128_P = 17,        // row-image kernel with four dedicated loader waves (k_conv_dma_h3p): 128 rows, eight matrix waves of 32 x 64
    CD_192_P = 18,        // ... 192 rows, six matrix waves of 64 x 64
    CD_256_P = 19,        // ... 256 rows, eight matrix waves of 64 x 64
    CD_128_N = 20,        // NARROW tiles, rows x 64 channels, loader waves (k_conv_dma_h3n): 128 rows
    CD_192_N = 21,        // ... 192 rows
    CD_256_N = 22         // ... 256 rows
};
template <int RH, int NB = 1, int MI = 2>
struct ConvDmaGeom {
    static constexpr int BM = 64 * RH, BN = 128 * NB, NW = 2 * RH * (2 / MI);
    static constexpr int A_BYTES = BM * 128, B_BYTES = 8 * BN * 16, BUF_BYTES = A_BYTES + B_BYTES;
    static constexpr int NPA = BM / 8, NPB = 16 * NB;               // 1 KB DMA pieces per step: A = 8 pixels x 128 B, B = 64 packed weight slots
    static constexpr int NA = NPA / NW, NBW = (NPB + NW - 1) / NW;   // A / B pieces per wave and step
    static constexpr int NPW = NA + NBW;                             // (RH = 3: 42 for 40, two waves repeat a B piece)
    static_assert(MI == 1 || MI == 2, "waves of 32 or 64 rows");
    static_assert(NPA % NW == 0, "every wave stages the same number of A pieces per step");
};

// NB = 2: 256 output channels per tile (a wave multiplies 64 rows x 128 channels = two 64-slot groups of the packed weights): 64 KB
// per K step for 256 MFMAs -- 31 bytes per clock and CU from L2 at the full MFMA rate, against 47 for the 256 x 128 tile and 62
// for 128 x 128: with every CU streaming, the L2 -> CU fabric is what these kernels run into first.
#ifdef TD_DMA_TRACE   // tools/conv_dma_trace.hip only: s_memtime stamps (shader cycles) of workgroups 0..3, every wave, the first 24 K steps:
// [wg][wave][step][0..3] = after the DMA issue, after the MFMAs, after the vmcnt wait, after the barrier
#define TD_DMA_STAMP(slot) do { if (blockIdx.x < 4 && step < 24 && lane == 0) \
    TD_DMA_TRACE[(((size_t)blockIdx.x * 8 + wave) * 24 + step) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TD_DMA_STAMP(slot) ((void)0)
#endif
template <int RH, int KS, bool OUT16, int NBUF, int NB = 1, int MI = 2>
TD_KERNEL void TD_LAUNCH_BOUNDS(256 * RH / MI, 1) k_conv_dma_h(ConvArgs p) {
    using G = ConvDmaGeom<RH, NB, MI>;
    static_assert(NB == 1 || (NB == 2 && RH == 4 && NBUF == 2 && MI == 2), "256-channel tiles: 256 rows, two LDS buffers");
    constexpr int NA = G::NA;
    constexpr int NJ = 2 * NB;                                      // 32-column accumulators per wave
    constexpr int BM = G::BM, NW = G::NW, NPW = G::NPW, NTAPS = KS * KS;
    static_assert(NBUF >= 2 && NBUF <= 4, "ring of 2, 3 or 4 LDS buffers");
    constexpr int INFLIGHT = (NBUF - 2) * G::NPW;                    // pieces of this wave that may still be in flight when the next step starts
    TD_DYN_LDS(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * G::BN;

    // ---- DMA geometry: A piece j of this wave = rows 8 (wave + NW j) .. + 7; lane l -> row + (l >> 3), LDS slot l & 7, which holds
    // the channels 8 kq .. 8 kq + 7 of the chunk with kq = slot ^ ((row >> 1) & 7) ------------------------------------------------
    unsigned a_off[NA], a_taps[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
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
    unsigned b_off[NPW - NA];
#pragma unroll
    for (int jb = 0; jb < NPW - NA; ++jb) {
        int pb = wave + NW * jb;
        if (pb >= G::NPB) pb -= NW;                                  // RH = 3: the two surplus slots repeat a piece (same bytes, same place)
        b_off[jb] = (unsigned)((pb / (2 * NB)) * p.CoutPad + n0 + (pb % (2 * NB)) * 64 + lane) * 16u;
    }

    // ---- DMA issue, one piece at a time.  An LDS-DMA instruction costs the issuing wave 100-150 cycles (tools/conv_dma_trace.hip: the
    // eight pieces of a step issued back to back right after the barrier took 700-1200 cycles, in both waves of every SIMD at once,
    // with the matrix pipe idle -- a third of the step).  So the pieces of step s + NBUF - 1 are issued BETWEEN the MFMA groups of
    // step s (NSLOT slots of four MFMAs), where the SIMD's other wave and the wave's own queued MFMAs cover them.
    int l_step = 0, l_chunk = 0, l_tap = 0;
    unsigned i_delta = 0u, i_wsoff = 0u;                              // of the step being issued
    bool i_live = true;                                               // false past the last step: the ring's surplus issues fetch nothing (zero fill)
    auto issue_begin = [&]() {
        i_live = l_step < p.nsteps;
        const int ky = l_tap / KS;
        const int dy = ky * p.dil, dx = (l_tap - ky * KS) * p.dil;
        i_delta = (unsigned)((dy * p.W + dx) * p.Cin + l_chunk * 64) * 2u;
        i_wsoff = (unsigned)(i_live ? l_step : 0) * w_step_bytes;
    };
    auto issue_piece = [&](int buf, int pc) {                         // pc: compile-time piece number of this wave, 0..NPW-1
        char* base = smem + buf * G::BUF_BYTES;
        if (pc < NA) {
            const bool ok = ((a_taps[pc] >> l_tap) & 1u) != 0u && i_live;
            td_buf_ld16_lds(in_buf, base + (wave + NW * pc) * 1024, ok ? a_off[pc] + i_delta : TD_BUF_OOB, 0u);
        } else {
            int pb = wave + NW * (pc - NA);
            if (pb >= G::NPB) pb -= NW;
            td_buf_ld16_lds(w_buf, base + G::A_BYTES + pb * 1024, i_live ? b_off[pc - NA] : TD_BUF_OOB, i_wsoff);
        }
    };
    auto issue_end = [&]() {
        ++l_step;
        if (++l_tap == NTAPS) { l_tap = 0; ++l_chunk; }
    };
    auto issue_all = [&](int buf) {
        issue_begin();
#pragma unroll
        for (int pc = 0; pc < NPW; ++pc) issue_piece(buf, pc);
        issue_end();
    };

    // ---- MFMA fragment addresses (bytes inside a buffer) -----------------------------------------------------------------
    unsigned a_rd[MI][4];                                             // [i][g]: row wm 32 MI + 32 i + l31, k-group 2 g + half, swizzled slot
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * 32 * MI + 32 * i + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) a_rd[i][g] = (unsigned)(row * 128 + (((2 * g + half) ^ ((row >> 1) & 7)) << 4));
    }
    constexpr int BKQ = G::BN * 16;                                   // bytes per k-group of the weight image
    const unsigned b_rd = (unsigned)(G::A_BYTES + half * BKQ + (wn * 64 * NB + l31) * 16);

    f32x16 acc[MI][NJ];                                               // [i][2 sb + nt]: rows 32 i .., packed slots wn 64 NB + 64 sb + 32 nt ..
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one K step on buffer `buf`, with the NPW pieces of the next issue spread over its 8 MFMA groups (ibuf < 0: nothing to issue)
    // One piece per group of MFMAs (8 groups per step; 256-channel tiles: four MFMAs per group, two otherwise).  Measured with the
    // in-kernel stamps (profiles/r03g_*): issue phase 700-1200 -> 200 cycles per step, 256 x 256 tile 150 -> 140 us per launch; packing
    // the pieces into the first four groups instead changed nothing for that tile and cost the 128 x 128 one 3 %.
    constexpr int NSLOT = 4 * MI;
    auto compute = [&](int buf, int ibuf) {
        const char* base = smem + buf * G::BUF_BYTES;
        f16x8 af[2][MI], bf[2][NJ];
        issue_begin();
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(base + a_rd[i][0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(base + b_rd + j * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const f16x8*>(base + a_rd[i][g + 1]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f16x8*>(base + b_rd + (g + 1) * 2 * BKQ + j * 512);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = td_mfma32_f16(af[g & 1][i], bf[g & 1][j], acc[i][j]);
                TD_SCHED_FENCE();
#pragma unroll
                for (int pc = 0; pc < NPW; ++pc)
                    if (pc % NSLOT == MI * g + i) issue_piece(ibuf, pc);
                TD_SCHED_FENCE();
            }
        }
        issue_end();
    };

    // ---- ring: the DMA of step s + NBUF - 1 is issued during the MFMAs of step s; a counted wait leaves the newest step(s) in
    // flight.  Order per step: compute (+ issue), wait for the NEXT step's pieces (this wave's), barrier (everyone's have landed, and
    // everyone is done reading the buffer the next issue overwrites).
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) issue_all(b);
    TD_WAIT_VM_PIECES(INFLIGHT);
    TD_BARRIER_RAW();
    int cb = 0, ib = NBUF - 1;                                        // buffer being multiplied / buffer being filled
    for (int step = 0; step < p.nsteps; ++step) {
        TD_DMA_STAMP(0);
        compute(cb, ib);
        TD_DMA_STAMP(1);
        TD_WAIT_VM_PIECES(INFLIGHT);
        TD_DMA_STAMP(2);
        TD_BARRIER_RAW();
        TD_DMA_STAMP(3);
        cb = cb + 1 == NBUF ? 0 : cb + 1;
        ib = ib + 1 == NBUF ? 0 : ib + 1;
    }
    TD_WAIT_VM_PIECES(0);                                             // the surplus pieces must not land in an LDS that has been handed on

    if constexpr (NB == 1) {
        td_store_acc_h<MI, 2, OUT16, true>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * 32 * MI, n0 + wn * 64, lane);
    } else if constexpr (MI == 2) {
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

// ---- 3x3, stride 1, "same" padding: the three taps of a kernel ROW read one LDS image ------------------------------------------------
// k_conv_dma_h stages the activation operand once per TAP: the same input pixels travel global -> LDS nine times per 64 channels.
// Here a "super-step" (64 input channels, one kernel row ky) stages ONE image: the input pixels
// of row offset (ky - 1) dil under the tile's BM consecutive output pixels, in a linear order WITH a horizontal halo of dil zero
// columns on either side of every image row (halo-linear index gs = iy' (W + 2 dil) + ix + dil); the tap kx of output pixel (oy, ox)
// is then slot (oy W' + ox - gs0) + kx dil of the same image, for all three kx.  A tile of BM pixels spans R <= (BM - 2) / W + 2 image
// rows, so the image has S = BM + 2 dil R slots instead of 3 BM: the activation traffic (L2 -> CU and LDS writes) of a 3x3 conv
// drops to a third plus the halo.  Slots are XOR-swizzled by (slot >> 1) & 7 like the rows of k_conv_dma_h; a fragment whose 32 rows
// straddle two image rows jumps by 2 dil slots there (measured: bank conflicts 6 % of the LDS cycles, 0 in k_conv_dma_h).
//   * two images (super-step u + 1 is staged during the steps kx = 0 and kx = 1 of super-step u: SH0 / SH1 pieces per wave), a ring
//     of NBB weight buffers as before; per step the weight pieces are issued first and the counted wait lets this step's image
//     pieces fly: vmcnt((NBB - 2) NBW + share(kx)).
//   * steps are unrolled by three so that the tap's fragment addresses are compile-time registers.
template <int RH, int NB, int MI, int NBB, int SH0, int SH1>
struct ConvDma3Geom {
    using G = ConvDmaGeom<RH, NB, MI>;
    static constexpr int NAP = SH0 + SH1;                            // image pieces per wave
    static constexpr int CAP = NAP * G::NW * 8;                      // image capacity in slots (pixels)
    static constexpr int IMG_BYTES = CAP * 128, LDS_BYTES = 2 * IMG_BYTES + NBB * G::B_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024 - 1024, "LDS budget");
    static_assert(NBB == 6 ? 3 * G::NBW + NAP <= 12 * MI : G::NBW + (SH0 > SH1 ? SH0 : SH1) <= 4 * MI, "one DMA piece per MFMA group");
};
template <int RH, int OUT16, int NB, int MI, int NBB, int SH0, int SH1>
TD_KERNEL void TD_LAUNCH_BOUNDS(256 * RH / MI, 1) k_conv_dma_h3(ConvArgs p) {
    using G = ConvDmaGeom<RH, NB, MI>;
    using G3 = ConvDma3Geom<RH, NB, MI, NBB, SH0, SH1>;
    constexpr int NJ = 2 * NB, BM = G::BM, NW = G::NW, NBW = G::NBW, NAP = G3::NAP;
    static_assert(NBB == 2 || NBB == 3 || NBB == 4 || NBB == 6, "two or three weight buffers; four = data lands a step early; six = one barrier per super-step");
    TD_DYN_LDS(smem);
    char* const wbase = smem + 2 * G3::IMG_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * G::BN;
    const int d = p.dil, Wh = p.W + 2 * d;
    const int oy0 = m0 / p.W, ox0 = m0 - oy0 * p.W;
    const int gs0 = oy0 * Wh + ox0;                                   // halo-linear index of image slot 0
    const int S = BM + 2 * d * ((BM - 2) / p.W + 2);                  // slots any tile can need

    // ---- image pieces: piece j of this wave = slots 8 (wave + NW j) .. + 7; lane -> slot + (lane >> 3), 16-byte LDS slot lane & 7 ------
    unsigned a_base[NAP], a_ok[NAP];                                  // source offset for ky = 1, chunk 0; bit ky: the row exists (and the column does)
#pragma unroll
    for (int j = 0; j < NAP; ++j) {
        const int sl = 8 * (wave + NW * j) + (lane >> 3);
        const int gs = gs0 + sl;
        const int r = gs / Wh, ix = gs - r * Wh - d;
        const int kq = (lane & 7) ^ ((sl >> 1) & 7);
        a_base[j] = (((unsigned)r * (unsigned)p.W + (unsigned)ix) * (unsigned)p.Cin + (unsigned)kq * 8u) * 2u;
        const bool xok = (unsigned)ix < (unsigned)p.W && sl < S;
        a_ok[j] = 0u;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a_ok[j] |= (xok && (unsigned)(r + (ky - 1) * d) < (unsigned)p.H) ? (1u << ky) : 0u;
    }
    const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 2u);
    const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
    const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;
    unsigned b_off[NBW];
#pragma unroll
    for (int jb = 0; jb < NBW; ++jb) {
        int pb = wave + NW * jb;
        if (pb >= G::NPB) pb -= NW;
        b_off[jb] = (unsigned)((pb / (2 * NB)) * p.CoutPad + n0 + (pb % (2 * NB)) * 64 + lane) * 16u;
    }
    const int nsuper = p.nsteps / 3;
    auto issue_image_piece = [&](int u, int j) {                      // piece j of the image of super-step u (chunk u / 3, kernel row u % 3)
        const int chunk = u / 3, ky = u - chunk * 3;
        const int delta = (((ky - 1) * d * p.W) * p.Cin + chunk * 64) * 2;
        const bool ok = u < nsuper && ((a_ok[j] >> ky) & 1u) != 0u;
        td_buf_ld16_lds(in_buf, smem + (u & 1) * G3::IMG_BYTES + (wave + NW * j) * 1024, ok ? a_base[j] + (unsigned)delta : TD_BUF_OOB, 0u);
    };
    auto issue_weight_piece = [&](int step, int buf, int jb) {       // piece jb of the weights of K step `step` into weight buffer `buf`
        int pb = wave + NW * jb;
        if (pb >= G::NPB) pb -= NW;
        const bool live = step < p.nsteps;
        td_buf_ld16_lds(w_buf, wbase + buf * G::B_BYTES + pb * 1024, live ? b_off[jb] : TD_BUF_OOB, (unsigned)(live ? step : 0) * w_step_bytes);
    };

    // ---- fragment addresses: a_rd[kx][i] = byte address of k-group `half` of this lane's row in the image, tap kx; k-group 2 g + half is
    // the same address with bits 5-6 XORed by g (the swizzle key touches bits 4-6 only) ------------------------------------------------
    unsigned a_rd[3][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * 32 * MI + 32 * i + l31;
        const int oy = m / p.W, ox = m - oy * p.W;
        const int sm = (oy - oy0) * Wh + ox - ox0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sl = sm + kx * d;
            a_rd[kx][i] = (unsigned)(sl * 128 + (((half) ^ ((sl >> 1) & 7)) << 4));
        }
    }
    constexpr int BKQ = G::BN * 16;
    const unsigned b_rd = (unsigned)(half * BKQ + (wn * 64 * NB + l31) * 16);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the MFMAs of one K step: tap KX of image `img` against the weights at `wb`; issue(slot) is called after every group of NJ MFMAs
    // (slot = MI g + i, 0 .. 4 MI - 1) so that the DMA pieces of later steps go out between them
    auto mma = [&](auto kx_tag, const char* img, const char* wb, auto&& issue) {
        constexpr int KX = decltype(kx_tag)::value;
        f16x8 af[2][MI], bf[2][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(img + a_rd[KX][i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + j * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const f16x8*>(img + (a_rd[KX][i] ^ (unsigned)((g + 1) << 5)));
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + (g + 1) * 2 * BKQ + j * 512);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = td_mfma32_f16(af[g & 1][i], bf[g & 1][j], acc[i][j]);
                TD_SCHED_FENCE();
                issue(MI * g + i);
                TD_SCHED_FENCE();
            }
        }
    };

    if constexpr (NBB == 6) {
        // ---- one barrier per SUPER-step (small tiles: a K step of 512 MFMA cycles paid ~450 cycles of barrier, wait and fragment
        // pipeline restart).  Weight buffers (u & 1) 3 + kx; the image and the three weight steps of super-step u + 1 are issued in the
        // first groups of super-step u (12 MI issue slots), everything is waited for at its end.
        auto tap = [&](auto kx_tag, int u) {
            constexpr int KX = decltype(kx_tag)::value;
            mma(kx_tag, smem + (u & 1) * G3::IMG_BYTES, wbase + ((u & 1) * 3 + KX) * G::B_BYTES, [&](int slot) {
                const int q = KX * 4 * MI + slot;                      // compile-time after unrolling
#pragma unroll
                for (int pc = 0; pc < 3 * NBW + NAP; ++pc)
                    if (pc == q) {
                        if (pc < 3 * NBW) issue_weight_piece(3 * (u + 1) + pc / NBW, ((u + 1) & 1) * 3 + pc / NBW, pc % NBW);
                        else issue_image_piece(u + 1, pc - 3 * NBW);
                    }
            });
        };
#pragma unroll
        for (int j = 0; j < NAP; ++j) issue_image_piece(0, j);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int jb = 0; jb < NBW; ++jb) issue_weight_piece(t, t, jb);
        TD_WAIT_VM_PIECES(0);
        TD_BARRIER_RAW();
        for (int u = 0; u < nsuper; ++u) {
            tap(std::integral_constant<int, 0>{}, u);
            tap(std::integral_constant<int, 1>{}, u);
            tap(std::integral_constant<int, 2>{}, u);
            TD_WAIT_VM_PIECES(0);
            TD_BARRIER_RAW();
        }
    } else if constexpr (NBB == 4) {
        // ---- data lands one step EARLY, so the fragment pipeline never drains at the barrier.  The barrier that ends step s also says
        // "everybody's pieces of step s + 2 have landed"; step s + 1 may therefore read the first fragments of step s + 2 BEFORE its own
        // end barrier, under its last MFMAs -- a step no longer starts with a cold ds_read after the barrier (measured on the per-step
        // form: MFMA pipes 39 % busy, waves 39 % waiting, LDS 29 % busy: nothing saturated, a dependency chain).  Weights: ring of four,
        // step s issues step s + 3.  Image of super-step u + 1: shares in the steps kx = 0 and kx = 1 of super-step u, complete at the end
        // of kx = 1 (in kx = 1 the image pieces go first so that the counted wait covers them and lets the newest weights fly).
        f16x8 af[2][MI], bf[2][NJ];                                   // fragment double buffer, live across steps
        auto frag0 = [&](auto kx_tag, int u, int step) {              // the k-group-0 fragments of (tap kx, image u & 1, weights step & 3) into slot 0
            constexpr int kx = decltype(kx_tag)::value;
            const char* img = smem + (u & 1) * G3::IMG_BYTES;
            const char* wb = wbase + (step & 3) * G::B_BYTES;
#pragma unroll
            for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(img + a_rd[kx][i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + j * 512);
        };
        auto compute = [&](auto kx_tag, int u, int step) {
            constexpr int KX = decltype(kx_tag)::value;
            constexpr int SHARE = KX == 0 ? SH0 : KX == 1 ? SH1 : 0, J0 = KX == 0 ? 0 : SH0;
            const char* img = smem + (u & 1) * G3::IMG_BYTES;
            const char* wb = wbase + (step & 3) * G::B_BYTES;
            const int ibuf = (step + 3) & 3;                           // the buffer of step - 1
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const f16x8*>(img + (a_rd[KX][i] ^ (unsigned)((g + 1) << 5)));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + (g + 1) * 2 * BKQ + j * 512);
                } else {
                    if constexpr (KX == 2) frag0(std::integral_constant<int, 0>{}, u + 1, step + 1);    // landed since the previous barrier
                    else frag0(std::integral_constant<int, KX + 1>{}, u, step + 1);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = td_mfma32_f16(af[g & 1][i], bf[g & 1][j], acc[i][j]);
                    TD_SCHED_FENCE();
                    const int slot = MI * g + i;
#pragma unroll
                    for (int pc = 0; pc < NBW + SHARE; ++pc)
                        if (pc == slot) {
                            const bool image_first = KX == 1;
                            const bool is_image = image_first ? pc < SHARE : pc >= NBW;
                            if (is_image) issue_image_piece(u + 1, J0 + (image_first ? pc : pc - NBW));
                            else issue_weight_piece(step + 3, ibuf, image_first ? pc - SHARE : pc);
                        }
                    TD_SCHED_FENCE();
                }
            }
            if constexpr (KX == 0) TD_WAIT_VM_PIECES(NBW + SH0); else TD_WAIT_VM_PIECES(NBW);
            TD_BARRIER_RAW();
        };
#pragma unroll
        for (int j = 0; j < NAP; ++j) issue_image_piece(0, j);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int jb = 0; jb < NBW; ++jb) issue_weight_piece(b, b, jb);
        TD_WAIT_VM_PIECES(NBW);                                       // image 0 and the weights of steps 0 and 1
        TD_BARRIER_RAW();
        frag0(std::integral_constant<int, 0>{}, 0, 0);
        for (int u = 0; u < nsuper; ++u) {
            compute(std::integral_constant<int, 0>{}, u, 3 * u);
            compute(std::integral_constant<int, 1>{}, u, 3 * u + 1);
            compute(std::integral_constant<int, 2>{}, u, 3 * u + 2);
        }
    } else {
        // one K step: this step issues the weights of step + NBB - 1 and, for KX < 2, a share of the next super-step's image
        auto compute = [&](auto kx_tag, int u, int step) {
            constexpr int KX = decltype(kx_tag)::value;
            constexpr int SHARE = KX == 0 ? SH0 : KX == 1 ? SH1 : 0, J0 = KX == 0 ? 0 : SH0;
            const int wbuf = NBB == 3 ? KX : ((u + KX) & 1);           // step = 3 u + KX
            const int ibuf = NBB == 3 ? (KX + 2) % 3 : (wbuf ^ 1);      // the buffer of step - 1, free since the last barrier
            mma(kx_tag, smem + (u & 1) * G3::IMG_BYTES, wbase + wbuf * G::B_BYTES, [&](int slot) {
#pragma unroll
                for (int pc = 0; pc < NBW + SHARE; ++pc)
                    if (pc == slot) {
                        if (pc < NBW) issue_weight_piece(step + NBB - 1, ibuf, pc);
                        else issue_image_piece(u + 1, J0 + pc - NBW);
                    }
            });
            TD_WAIT_VM_PIECES((NBB - 2) * NBW + SHARE);
            TD_BARRIER_RAW();
        };
        // prologue: the first image, the first NBB - 1 weight steps
#pragma unroll
        for (int j = 0; j < NAP; ++j) issue_image_piece(0, j);
#pragma unroll
        for (int b = 0; b < NBB - 1; ++b)
#pragma unroll
            for (int jb = 0; jb < NBW; ++jb) issue_weight_piece(b, b, jb);
        TD_WAIT_VM_PIECES((NBB - 2) * NBW);
        TD_BARRIER_RAW();
        for (int u = 0; u < nsuper; ++u) {
            compute(std::integral_constant<int, 0>{}, u, 3 * u);
            compute(std::integral_constant<int, 1>{}, u, 3 * u + 1);
            compute(std::integral_constant<int, 2>{}, u, 3 * u + 2);
        }
    }
    TD_WAIT_VM_PIECES(0);                                             // the surplus (zero-fill) pieces must not land in an LDS that has been handed on

    if constexpr (NB == 1) {
        td_store_acc_h<MI, 2, OUT16 != 0, true>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * 32 * MI, n0 + wn * 64, lane);
    } else if constexpr (MI == 2) {
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) {
            f32x16 part[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) part[i][nt] = acc[i][2 * sb + nt];
            td_store_acc_h<2, 2, OUT16 != 0, true>(part, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * 64, n0 + wn * 64 * NB + sb * 64, lane);
        }
    }
}
// slots an image needs for BM-pixel tiles of a W-wide map at dilation d (the kernel computes the same S)
static inline int conv_dma3_slots(int BM, int W, int d) { return BM + 2 * d * ((BM - 2) / W + 2); }
template <int RH, int NB, int MI, int NBB, int SH0, int SH1>
static inline bool conv_launch_dma3_t(const ConvArgs& a, bool out16, hipStream_t s) {
    using G = ConvDmaGeom<RH, NB, MI>;
    using G3 = ConvDma3Geom<RH, NB, MI, NBB, SH0, SH1>;
    if (conv_dma3_slots(G::BM, a.W, a.dil) > G3::CAP) return false;
    const int grid = ((a.M + G::BM - 1) / G::BM) * a.tiles_n;
    if (out16) TD_LAUNCH((k_conv_dma_h3<RH, 1, NB, MI, NBB, SH0, SH1>), dim3(grid), dim3(64 * G::NW), G3::LDS_BYTES, s, a);
    else TD_LAUNCH((k_conv_dma_h3<RH, 0, NB, MI, NBB, SH0, SH1>), dim3(grid), dim3(64 * G::NW), G3::LDS_BYTES, s, a);
    return true;
}
// the row-image kernel for the tile code rh of conv_launch_dma, where the conv qualifies (3x3, stride 1, padding = dilation, the halo fits);
// false = not launched, use conv_launch_dma
static inline bool conv_launch_dma3(ConvArgs a, int rh, int KS, bool out16, hipStream_t s) {
    if (KS != 3 || a.stride != 1 || a.pad != a.dil || a.Wo != a.W || a.nsteps % 3) return false;
    a.tiles_n = a.CoutPad / (rh == CD_256x256 ? 256 : 128);
    switch (rh) {
        case CD_256x256: return conv_launch_dma3_t<4, 2, 2, 2, 3, 2>(a, out16, s);
        // 256 x 128 and 192 x 128: the early-landing form where its (smaller) image buffer holds the halo, else one barrier per K step
        case CD_256: return conv_launch_dma3_t<4, 1, 2, 4, 3, 2>(a, out16, s) || conv_launch_dma3_t<4, 1, 2, 3, 3, 3>(a, out16, s);
        case CD_192: return conv_launch_dma3_t<3, 1, 2, 4, 3, 3>(a, out16, s) || conv_launch_dma3_t<3, 1, 2, 3, 3, 3>(a, out16, s);
        // 128 rows, eight waves, where the grid leaves one workgroup per CU: per-step, super-step and early forms all take 22.5-23.3 us
        // at 720x960 / 256 channels (profiles/r03w_*); the plain per-step form (largest image buffer) is the default
        case CD_128:
            if ((long)((a.M + 127) / 128) * a.tiles_n > TD_CUS) return false;           // larger grids: two workgroups per CU on the tap-by-tap kernel
            [[fallthrough]];
        case CD_128_8W: case CD_128_STEP: return conv_launch_dma3_t<2, 1, 1, 3, 2, 2>(a, out16, s);
        // single forms, for the probes and tests
        case CD_256_EARLY: return conv_launch_dma3_t<4, 1, 2, 4, 3, 2>(a, out16, s);
        case CD_192_EARLY: return conv_launch_dma3_t<3, 1, 2, 4, 3, 3>(a, out16, s);
        case CD_192_STEP: return conv_launch_dma3_t<3, 1, 2, 3, 3, 3>(a, out16, s);
        case CD_192_SUPER: return conv_launch_dma3_t<3, 1, 2, 6, 3, 2>(a, out16, s);
        case CD_128_EARLY: return conv_launch_dma3_t<2, 1, 1, 4, 2, 2>(a, out16, s);
        case CD_128_SUPER: return conv_launch_dma3_t<2, 1, 1, 6, 2, 1>(a, out16, s);
        default: return false;
    }
}

// ---- the row-image kernel with DEDICATED LOADER WAVES (round 4) ------------------------------------------------------------------------------
// Across the four tile shapes of k_conv_dma_h / _h3 a K step takes (pieces x 16 cycles) + (MFMA cycles of a SIMD), i.e. the LDS-DMA
// time of the step's bytes at the CU's 64 B/clk PLUS its matrix time, as if the two never overlapped (128 x 128: 512 + 512 -> 0.56 us
// measured 0.6; 192 x 128: 640 + 768 -> 0.81; 256 x 128: 768 + 1024 -> 0.98; profiles/r03w_*, r04c_*).  They do not overlap because the
// wave that issues a `buffer_load ... lds` waits 100-185 cycles on it (MI355X_MICROARCH.md: "LDS-DMA piece issue cost") -- in the middle
// of its own MFMA stream: every matrix wave spends as long issuing DMA as multiplying, and two such waves per SIMD interleave only
// half of that away.  Here NP = 4 extra waves (one per SIMD) do NOTHING but issue the step's pieces back to back, wait for the
// previous step's with a counted vmcnt and meet the barrier; the matrix waves read fragments and multiply and never touch vector
// memory inside the K loop.  Same images, same weights ring (three buffers), same products in the same order as k_conv_dma_h3 <.., 3, ..>:
// bit-identical results.  Image pieces: IP per loader and super-step (capacity IP * NP * 8 slots), issued in the steps kx = 0 and 1.
template <int RH, int MI, int NP, int IP>
struct ConvDmaPGeom {
    using G = ConvDmaGeom<RH, 1, MI>;
    static constexpr int NWC = G::NW;                               // matrix (consumer) waves
    static constexpr int CAP = IP * NP * 8;                         // image capacity in slots (pixels)
    static constexpr int IMG_BYTES = CAP * 128, NBB = 3, LDS_BYTES = 2 * IMG_BYTES + NBB * G::B_BYTES;
    static constexpr int WPP = G::NPB / NP;                         // weight pieces per loader and step
    static constexpr int SH0 = (IP + 1) / 2, SH1 = IP - SH0;
    static_assert(G::NPB % NP == 0, "every loader stages the same number of weight pieces");
    static_assert(LDS_BYTES <= 160 * 1024 - 1024, "LDS budget");
};
#ifdef TD_P_TRACE      // tools/conv_h3p_trace.hip only: s_memtime stamps of workgroups 0..3, every wave (matrix and loader), the first 24 K steps:
// matrix waves [0] step start, [1] first k-group's MFMAs issued, [2] all MFMAs issued, [3] after the barrier;
// loader waves [0] step start, [1] pieces issued, [2] after the counted wait, [3] after the barrier
#define TD_P_STAMP(st_, slot) do { if (blockIdx.x < 4 && (st_) < 24 && lane == 0) \
    TD_P_TRACE[(((size_t)blockIdx.x * 16 + wave) * 24 + (st_)) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TD_P_STAMP(st_, slot) ((void)0)
#endif
template <int RH, int OUT16, int MI, int NP, int IP>
TD_KERNEL void TD_LAUNCH_BOUNDS(64 * (2 * RH * (2 / MI) + NP), 1) k_conv_dma_h3p(ConvArgs p) {
    using G = ConvDmaGeom<RH, 1, MI>;
    using GP = ConvDmaPGeom<RH, MI, NP, IP>;
    constexpr int NJ = 2, BM = G::BM, NWC = GP::NWC, WPP = GP::WPP;
    TD_DYN_LDS(smem);
    char* const wbase = smem + 2 * GP::IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * G::BN;
    const int d = p.dil, Wh = p.W + 2 * d;
    const int oy0 = m0 / p.W, ox0 = m0 - oy0 * p.W;
    const int gs0 = oy0 * Wh + ox0;                                   // halo-linear index of image slot 0
    const int S = BM + 2 * d * ((BM - 2) / p.W + 2);                  // slots any tile can need
    const int nsuper = p.nsteps / 3;

    if (wave >= NWC) {
        // =========================== loader wave pw: image pieces pw + NP j, weight pieces pw + NP jb ===========================
        const int pw = wave - NWC;
        unsigned a_base[IP], a_ok[IP];
#pragma unroll
        for (int j = 0; j < IP; ++j) {
            const int sl = 8 * (pw + NP * j) + (lane >> 3);
            const int gs = gs0 + sl;
            const int r = gs / Wh, ix = gs - r * Wh - d;
            const int kq = (lane & 7) ^ ((sl >> 1) & 7);
            a_base[j] = (((unsigned)r * (unsigned)p.W + (unsigned)ix) * (unsigned)p.Cin + (unsigned)kq * 8u) * 2u;
            const bool xok = (unsigned)ix < (unsigned)p.W && sl < S;
            a_ok[j] = 0u;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) a_ok[j] |= (xok && (unsigned)(r + (ky - 1) * d) < (unsigned)p.H) ? (1u << ky) : 0u;
        }
        const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 2u);
        const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
        const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;
        unsigned b_off[WPP];
#pragma unroll
        for (int jb = 0; jb < WPP; ++jb) {
            const int pb = pw + NP * jb;                                // piece pb = 2 kq + q: 64 consecutive packed slots of k-group kq
            b_off[jb] = (unsigned)((pb / 2) * p.CoutPad + n0 + (pb % 2) * 64 + lane) * 16u;
        }
        auto issue_image_piece = [&](int u, int j) {                  // u: super-step = (64-channel chunk, kernel row)
            const int chunk = u / 3, ky = u - chunk * 3;
            const int delta = (((ky - 1) * d * p.W) * p.Cin + chunk * 64) * 2;
            const bool ok = u < nsuper && ((a_ok[j] >> ky) & 1u) != 0u;
            td_buf_ld16_lds(in_buf, smem + (u & 1) * GP::IMG_BYTES + (pw + NP * j) * 1024, ok ? a_base[j] + (unsigned)delta : TD_BUF_OOB, 0u);
        };
        auto issue_weights = [&](int step, int buf) {
            const bool live = step < p.nsteps;
#pragma unroll
            for (int jb = 0; jb < WPP; ++jb)
                td_buf_ld16_lds(w_buf, wbase + buf * G::B_BYTES + (pw + NP * jb) * 1024, live ? b_off[jb] : TD_BUF_OOB, (unsigned)(live ? step : 0) * w_step_bytes);
        };
#pragma unroll
        for (int j = 0; j < IP; ++j) issue_image_piece(0, j);
        issue_weights(0, 0);
        issue_weights(1, 1);
        TD_WAIT_VM_PIECES(WPP);                                       // image 0 and the weights of step 0 (step 1's may fly)
        TD_BARRIER_RAW();
        for (int u = 0; u < nsuper; ++u) {
            // kx = 0: weights of step 3 u + 2 into the buffer step 3 u - 1 left, the first share of the next image
            TD_P_STAMP(3 * u, 0);
            issue_weights(3 * u + 2, 2);
#pragma unroll
            for (int j = 0; j < GP::SH0; ++j) issue_image_piece(u + 1, j);
            TD_P_STAMP(3 * u, 1);
            TD_WAIT_VM_PIECES(WPP + GP::SH0);                         // the weights of step 3 u + 1 have landed; this step's issues may fly
            TD_P_STAMP(3 * u, 2);
            TD_BARRIER_RAW();
            TD_P_STAMP(3 * u, 3);
            TD_P_STAMP(3 * u + 1, 0);
            issue_weights(3 * u + 3, 0);
#pragma unroll
            for (int j = GP::SH0; j < IP; ++j) issue_image_piece(u + 1, j);
            TD_P_STAMP(3 * u + 1, 1);
            TD_WAIT_VM_PIECES(WPP + GP::SH1);
            TD_P_STAMP(3 * u + 1, 2);
            TD_BARRIER_RAW();
            TD_P_STAMP(3 * u + 1, 3);
            TD_P_STAMP(3 * u + 2, 0);
            issue_weights(3 * u + 4, 1);
            TD_P_STAMP(3 * u + 2, 1);
            TD_WAIT_VM_PIECES(WPP);                                   // the next image (both shares) and the weights of step 3 u + 3
            TD_P_STAMP(3 * u + 2, 2);
            TD_BARRIER_RAW();
            TD_P_STAMP(3 * u + 2, 3);
        }
        TD_WAIT_VM_PIECES(0);                                         // the surplus (zero-fill) pieces must not land in an LDS that has been handed on
        return;
    }

    // =========================================== matrix wave: fragments and MFMAs only ===========================================
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    unsigned a_rd[3][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * 32 * MI + 32 * i + l31;
        const int oy = m / p.W, ox = m - oy * p.W;
        const int sm = (oy - oy0) * Wh + ox - ox0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sl = sm + kx * d;
            a_rd[kx][i] = (unsigned)(sl * 128 + (((half) ^ ((sl >> 1) & 7)) << 4));
        }
    }
    constexpr int BKQ = G::BN * 16;
    const unsigned b_rd = (unsigned)(half * BKQ + (wn * 64 + l31) * 16);
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma = [&](auto kx_tag, const char* img, const char* wb, int step) {
        constexpr int KX = decltype(kx_tag)::value;
        (void)step;
        f16x8 af[2][MI], bf[2][NJ];
        TD_P_STAMP(step, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const f16x8*>(img + a_rd[KX][i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + j * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const f16x8*>(img + (a_rd[KX][i] ^ (unsigned)((g + 1) << 5)));
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + (g + 1) * 2 * BKQ + j * 512);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = td_mfma32_f16(af[g & 1][i], bf[g & 1][j], acc[i][j]);
            if (g == 0) TD_P_STAMP(step, 1);
        }
        TD_P_STAMP(step, 2);
    };
    TD_BARRIER_RAW();                                                  // the loaders' prologue
    for (int u = 0; u < nsuper; ++u) {
        const char* img = smem + (u & 1) * GP::IMG_BYTES;
        mma(std::integral_constant<int, 0>{}, img, wbase, 3 * u);
        TD_BARRIER_RAW();
        TD_P_STAMP(3 * u, 3);
        mma(std::integral_constant<int, 1>{}, img, wbase + G::B_BYTES, 3 * u + 1);
        TD_BARRIER_RAW();
        TD_P_STAMP(3 * u + 1, 3);
        mma(std::integral_constant<int, 2>{}, img, wbase + 2 * G::B_BYTES, 3 * u + 2);
        TD_BARRIER_RAW();
        TD_P_STAMP(3 * u + 2, 3);
    }
    td_store_acc_h<MI, 2, OUT16 != 0, true>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wm * 32 * MI, n0 + wn * 64, lane);
}
template <int RH, int MI, int NP, int IP>
static inline bool conv_launch_dma3p_t(const ConvArgs& a, bool out16, hipStream_t s) {
    using G = ConvDmaGeom<RH, 1, MI>;
    using GP = ConvDmaPGeom<RH, MI, NP, IP>;
    if (conv_dma3_slots(G::BM, a.W, a.dil) > GP::CAP) return false;
    const int grid = ((a.M + G::BM - 1) / G::BM) * a.tiles_n;
    if (out16) TD_LAUNCH((k_conv_dma_h3p<RH, 1, MI, NP, IP>), dim3(grid), dim3(64 * (GP::NWC + NP)), GP::LDS_BYTES, s, a);
    else TD_LAUNCH((k_conv_dma_h3p<RH, 0, MI, NP, IP>), dim3(grid), dim3(64 * (GP::NWC + NP)), GP::LDS_BYTES, s, a);
    return true;
}
// rh: CD_128_P / CD_192_P / CD_256_P (x 128 channels); the smallest image buffer that holds the halo.  false = not launched.
static inline bool conv_launch_dma3p(ConvArgs a, int rh, int KS, bool out16, hipStream_t s) {
    if (KS != 3 || a.stride != 1 || a.pad != a.dil || a.Wo != a.W || a.nsteps % 3) return false;
    a.tiles_n = a.CoutPad / 128;
    switch (rh) {
        case CD_128_P: return conv_launch_dma3p_t<2, 1, 4, 5>(a, out16, s) || conv_launch_dma3p_t<2, 1, 4, 6>(a, out16, s) || conv_launch_dma3p_t<2, 1, 4, 8>(a, out16, s);
        case CD_192_P: return conv_launch_dma3p_t<3, 2, 4, 7>(a, out16, s) || conv_launch_dma3p_t<3, 2, 4, 9>(a, out16, s);
        case CD_256_P: return conv_launch_dma3p_t<4, 2, 4, 9>(a, out16, s) || conv_launch_dma3p_t<4, 2, 4, 11>(a, out16, s);
        default: return false;
    }
}

// ---- NARROW tiles: 64 RH rows x 64 channels, loader waves (round 4) ---------------------------------------------------------------------------
// Over the four tile shapes a K step of these kernels takes  bytes fetched / 100 GB/s  +  MFMA cycles / 1.42 GHz  (720x960 and 1024x2048,
// profiles/r04d_*, r04r_*: 128 x 128 0.58 us, 192 x 128 0.79, 256 x 128 1.11, 256 x 256 1.82): the CU's loads and its MFMAs add up instead of
// overlapping, whatever the synchronisation (per step, per super-step, LDS flags), the issuing waves (matrix, four or eight loaders) or the
// number of workgroups per CU.  On a map of 10800 pixels the bytes are the larger term, and most of them are WEIGHTS: 16 KB per step for a
// 128-column tile against ~6 KB of activations (row images).  A tile half as wide and 1.5x as tall -- 192 x 64: 57 x 4 = 228 workgroups for
// a 256-channel layer, 128 x 64: 85 x 2 = 170 for a 128-channel one -- fetches 8 + 9 = 17 KB (8 + 6 = 14) per step instead of 22.
// Same structure as k_conv_dma_h3p (four loader waves, one barrier per step, three weight buffers, two images); matrix waves of 32 x 64
// stacked along the rows only.  A 64-column group of the packed weights is a wave column of the 128-wide packing: same products, same
// order, bit-identical results.
// W32 = matrix waves = 32-row blocks of the tile: 4 / 6 / 8 = 128 / 192 / 256 rows.  (Round 5 also ran 2 / 3 = 64 / 96 rows -- grids of more than
// one workgroup per CU on a 10^4-pixel map -- and the 512-channel convs on these tiles: bit-identical, 1 - 7 % SLOWER in the 720x960 frame
// (profiles/r05c_*): two co-resident workgroups fetch twice the weights through the same CU and that, not latency, is the limit.  Removed.)
template <int W32, int NP, int IP>
struct ConvDmaNGeom {
    static constexpr int BM = 32 * W32, BN = 64, NWC = W32;          // matrix waves: 32 rows x 64 channels each
    static constexpr int B_BYTES = 8 * BN * 16;                       // 8 KB per K step
    static constexpr int CAP = IP * NP * 8, IMG_BYTES = CAP * 128, LDS_BYTES = 2 * IMG_BYTES + 3 * B_BYTES;
    static constexpr int WPP = 8 / NP, SH0 = (IP + 1) / 2, SH1 = IP - SH0;
    static_assert(8 % NP == 0, "every loader stages the same number of weight pieces");
    static_assert(LDS_BYTES <= 160 * 1024 - 1024, "LDS budget");
};
template <int W32, int OUT16, int NP, int IP>
TD_KERNEL void TD_LAUNCH_BOUNDS(64 * (W32 + NP), 1) k_conv_dma_h3n(ConvArgs p) {
    using GN = ConvDmaNGeom<W32, NP, IP>;
    constexpr int NJ = 2, BM = GN::BM, NWC = GN::NWC, WPP = GN::WPP;
    TD_DYN_LDS(smem);
    char* const wbase = smem + 2 * GN::IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = td_wave();
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * GN::BN;
    const int d = p.dil, Wh = p.W + 2 * d;
    const int oy0 = m0 / p.W, ox0 = m0 - oy0 * p.W;
    const int gs0 = oy0 * Wh + ox0;
    const int S = BM + 2 * d * ((BM - 2) / p.W + 2);
    const int nsuper = p.nsteps / 3;

    if (wave >= NWC) {
        const int pw = wave - NWC;
        unsigned a_base[IP], a_ok[IP];
#pragma unroll
        for (int j = 0; j < IP; ++j) {
            const int sl = 8 * (pw + NP * j) + (lane >> 3);
            const int gs = gs0 + sl;
            const int r = gs / Wh, ix = gs - r * Wh - d;
            const int kq = (lane & 7) ^ ((sl >> 1) & 7);
            a_base[j] = (((unsigned)r * (unsigned)p.W + (unsigned)ix) * (unsigned)p.Cin + (unsigned)kq * 8u) * 2u;
            const bool xok = (unsigned)ix < (unsigned)p.W && sl < S;
            a_ok[j] = 0u;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) a_ok[j] |= (xok && (unsigned)(r + (ky - 1) * d) < (unsigned)p.H) ? (1u << ky) : 0u;
        }
        const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 2u);
        const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
        const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;
        unsigned b_off[WPP];
#pragma unroll
        for (int jb = 0; jb < WPP; ++jb) b_off[jb] = (unsigned)((pw + NP * jb) * p.CoutPad + n0 + lane) * 16u;   // piece = k-group kq: 64 packed slots
        auto issue_image_piece = [&](int u, int j) {
            const int chunk = u / 3, ky = u - chunk * 3;
            const int delta = (((ky - 1) * d * p.W) * p.Cin + chunk * 64) * 2;
            const bool ok = u < nsuper && ((a_ok[j] >> ky) & 1u) != 0u;
            td_buf_ld16_lds(in_buf, smem + (u & 1) * GN::IMG_BYTES + (pw + NP * j) * 1024, ok ? a_base[j] + (unsigned)delta : TD_BUF_OOB, 0u);
        };
        auto issue_weights = [&](int step, int buf) {
            const bool live = step < p.nsteps;
#pragma unroll
            for (int jb = 0; jb < WPP; ++jb)
                td_buf_ld16_lds(w_buf, wbase + buf * GN::B_BYTES + (pw + NP * jb) * 1024, live ? b_off[jb] : TD_BUF_OOB, (unsigned)(live ? step : 0) * w_step_bytes);
        };
#pragma unroll
        for (int j = 0; j < IP; ++j) issue_image_piece(0, j);
        issue_weights(0, 0);
        issue_weights(1, 1);
        TD_WAIT_VM_PIECES(WPP);
        TD_BARRIER_RAW();
        for (int u = 0; u < nsuper; ++u) {
            issue_weights(3 * u + 2, 2);
#pragma unroll
            for (int j = 0; j < GN::SH0; ++j) issue_image_piece(u + 1, j);
            TD_WAIT_VM_PIECES(WPP + GN::SH0);
            TD_BARRIER_RAW();
            issue_weights(3 * u + 3, 0);
#pragma unroll
            for (int j = GN::SH0; j < IP; ++j) issue_image_piece(u + 1, j);
            TD_WAIT_VM_PIECES(WPP + GN::SH1);
            TD_BARRIER_RAW();
            issue_weights(3 * u + 4, 1);
            TD_WAIT_VM_PIECES(WPP);
            TD_BARRIER_RAW();
        }
        TD_WAIT_VM_PIECES(0);
        return;
    }

    const int half = lane >> 5, l31 = lane & 31;
    unsigned a_rd[3];
    {
        const int m = m0 + wave * 32 + l31;
        const int oy = m / p.W, ox = m - oy * p.W;
        const int sm = (oy - oy0) * Wh + ox - ox0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sl = sm + kx * d;
            a_rd[kx] = (unsigned)(sl * 128 + (((half) ^ ((sl >> 1) & 7)) << 4));
        }
    }
    constexpr int BKQ = GN::BN * 16;
    const unsigned b_rd = (unsigned)(half * BKQ + l31 * 16);
    f32x16 acc[1][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    auto mma = [&](auto kx_tag, const char* img, const char* wb) {
        constexpr int KX = decltype(kx_tag)::value;
        f16x8 af[2], bf[2][NJ];
        af[0] = *reinterpret_cast<const f16x8*>(img + a_rd[KX]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + j * 512);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
                af[(g + 1) & 1] = *reinterpret_cast<const f16x8*>(img + (a_rd[KX] ^ (unsigned)((g + 1) << 5)));
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const f16x8*>(wb + b_rd + (g + 1) * 2 * BKQ + j * 512);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[0][j] = td_mfma32_f16(af[g & 1], bf[g & 1][j], acc[0][j]);
        }
    };
    TD_BARRIER_RAW();
    for (int u = 0; u < nsuper; ++u) {
        const char* img = smem + (u & 1) * GN::IMG_BYTES;
        mma(std::integral_constant<int, 0>{}, img, wbase);
        TD_BARRIER_RAW();
        mma(std::integral_constant<int, 1>{}, img, wbase + GN::B_BYTES);
        TD_BARRIER_RAW();
        mma(std::integral_constant<int, 2>{}, img, wbase + 2 * GN::B_BYTES);
        TD_BARRIER_RAW();
    }
    td_store_acc_h<1, 2, OUT16 != 0, true>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wave * 32, n0, lane);
}
template <int W32, int NP, int IP>
static inline bool conv_launch_dma3n_t(const ConvArgs& a, bool out16, hipStream_t s) {
    using GN = ConvDmaNGeom<W32, NP, IP>;
    if (conv_dma3_slots(GN::BM, a.W, a.dil) > GN::CAP) return false;
    const int grid = ((a.M + GN::BM - 1) / GN::BM) * a.tiles_n;
    if (out16) TD_LAUNCH((k_conv_dma_h3n<W32, 1, NP, IP>), dim3(grid), dim3(64 * (GN::NWC + NP)), GN::LDS_BYTES, s, a);
    else TD_LAUNCH((k_conv_dma_h3n<W32, 0, NP, IP>), dim3(grid), dim3(64 * (GN::NWC + NP)), GN::LDS_BYTES, s, a);
    return true;
}
// rh: CD_128_N / CD_192_N / CD_256_N (rows x 64 channels); false = not launched
static inline bool conv_launch_dma3n(ConvArgs a, int rh, int KS, bool out16, hipStream_t s) {
    if (KS != 3 || a.stride != 1 || a.pad != a.dil || a.Wo != a.W || a.nsteps % 3) return false;
    a.tiles_n = (a.Cout + 63) / 64;                                   // (not CoutPad / 64: a 64-channel conv packed 128 wide has ONE column of work)
    switch (rh) {
        case CD_128_N: return conv_launch_dma3n_t<4, 4, 5>(a, out16, s) || conv_launch_dma3n_t<4, 4, 6>(a, out16, s) || conv_launch_dma3n_t<4, 4, 8>(a, out16, s);
        case CD_192_N: return conv_launch_dma3n_t<6, 4, 7>(a, out16, s) || conv_launch_dma3n_t<6, 4, 9>(a, out16, s);
        case CD_256_N: return conv_launch_dma3n_t<8, 4, 9>(a, out16, s) || conv_launch_dma3n_t<8, 4, 11>(a, out16, s);
        default: return false;
    }
}

// (Rounds 3-4 carried k_conv_dma_w64 here: 64 -> 64 channels on persistent workgroups with the weights resident in LDS -- 152 KB of LDS,
// one wave per SIMD; measured no faster than the per-tile kernel (22.3 vs 22.9 us at 1024x2048, 13.3 vs 13.0 at 720x960, profiles/r03x_*).
// Removed in round 5; last commit 78dfa5a.)

// Tile for an output of M pixels x Cout channels: CD_256 / CD_192 / CD_128 (256 / 192 / 128 rows x 128 channels), CD_256x256 (needs CoutPad
// % 256 == 0: the caller says so), or CD_NONE = leave the conv on the register-staged kernel with its 64 x 128 tiles (many small workgroups).
// Cost = what the busiest CU has to do: ceil(tiles / 256 CUs) tiles of rh x nb units, divided by the tile's relative throughput --
// measured on MI355X at 1024x2048 (profiles/r03d_*): 256 x 256 1070 TFLOP/s, 256 x 128 950, the register-staged 128 x 128 930; the
// smaller ones estimated from their bytes per MFMA.  At 720x960 (10800 pixels) this sends the 128-channel layers to the old 64 x 128
// tiles (169 workgroups instead of 57), the 256-channel ones to 128 x 128 (170) and the 512-channel ones to 192 x 128 (228).
static inline int conv_dma_pick_rh(long M, int Cout, bool allow256 = true) {
    static const struct { int code, rows, nb; double eff; } cand[5] = {{CD_256, 4, 1, 1.00}, {CD_192, 3, 1, 0.95}, {CD_128, 2, 1, 0.85}, {CD_256x256, 4, 2, 1.13}, {CD_NONE, 1, 1, 0.60}};
    int best = CD_256;
    double best_cost = 0.0;
    for (int i = 0; i < 5; ++i) {
        if (cand[i].nb == 2 && (!allow256 || Cout % 256)) continue;
        const long tn = (Cout + 128 * cand[i].nb - 1) / (128 * cand[i].nb);
        const long tiles = ((M + 64 * cand[i].rows - 1) / (64 * cand[i].rows)) * tn;
        const double cost = (double)((tiles + TD_CUS - 1) / TD_CUS) * cand[i].rows * cand[i].nb / cand[i].eff;
        if (i == 0 || cost < best_cost) { best_cost = cost; best = cand[i].code; }
    }
    return best;
}
static inline bool conv_dma_supports(int Cin, int Cout, int KS, ConvTile tile) {
    const ConvTileDims d = conv_tile_dims(tile);
    return Cin % 64 == 0 && Cout >= 64 && d.BN == 128 && d.WGN == 2 && (KS == 1 || KS == 3);   // Cout < 128: a half-empty tile (the caller's tile decides)
}

template <int RH, int NBUF, int NB, int MI = 2>
static inline void conv_launch_dma_t(const ConvArgs& a, int KS, bool out16, hipStream_t s) {
    using G = ConvDmaGeom<RH, NB, MI>;
    const int grid = ((a.M + G::BM - 1) / G::BM) * a.tiles_n;
    const int lds = NBUF * G::BUF_BYTES;
    if (KS == 3 && out16) TD_LAUNCH((k_conv_dma_h<RH, 3, true, NBUF, NB, MI>), dim3(grid), dim3(64 * G::NW), lds, s, a);
    else if (KS == 3) TD_LAUNCH((k_conv_dma_h<RH, 3, false, NBUF, NB, MI>), dim3(grid), dim3(64 * G::NW), lds, s, a);
    else if (out16) TD_LAUNCH((k_conv_dma_h<RH, 1, true, NBUF, NB, MI>), dim3(grid), dim3(64 * G::NW), lds, s, a);
    else TD_LAUNCH((k_conv_dma_h<RH, 1, false, NBUF, NB, MI>), dim3(grid), dim3(64 * G::NW), lds, s, a);
}
// CD_256 / CD_192: three LDS buffers, one workgroup per CU; CD_128: see below; CD_256x256: two buffers, one workgroup per CU.
// 128 x 128 has two forms: two buffers (64 KB: two workgroups per CU cover each other's waits) when the grid has more workgroups than
// CUs, and a ring of FOUR (128 KB) when it has not -- a lone workgroup of four waves issues the last piece of step s + 1 at the end of
// step s and then waits for it: every step paid a full memory latency (measured at 720x960, 256 channels, 170 workgroups: 36 steps in
// 35.6 us = 1 us per step for 0.25 us of MFMAs).  The four-buffer form runs as EIGHT waves of 32 x 64 (two per SIMD; 3-5 % faster than
// four of 64 x 64, tools/conv_h_ring_probe.sh).  A step still takes ~1000 cycles for 512 of MFMAs whatever the wave shape, the barrier
// placement or the activation bytes (pipes 39 % busy, LDS 29 %, waves waiting 39 %) -- a dependency chain that one
// workgroup per CU cannot overlap with anything.
static inline void conv_launch_dma(ConvArgs a, int rh, int KS, bool out16, hipStream_t s) {
    if (rh == CD_128_SUPER || rh == CD_128_STEP || rh == CD_128_EARLY) rh = CD_128_8W;   // forms of the row-image kernel: the same tile here
    if (rh == CD_192_SUPER || rh == CD_192_STEP || rh == CD_192_EARLY) rh = CD_192;
    if (rh == CD_256_EARLY) rh = CD_256;
    a.tiles_n = a.CoutPad / (rh == CD_256x256 ? 256 : 128);
    const bool one_per_cu = (long)((a.M + 127) / 128) * a.tiles_n <= TD_CUS;
    if (rh == CD_256x256) conv_launch_dma_t<4, 2, 2>(a, KS, out16, s);
    else if (rh == CD_256) conv_launch_dma_t<4, 3, 1>(a, KS, out16, s);
    else if (rh == CD_192) conv_launch_dma_t<3, 3, 1>(a, KS, out16, s);
    else if (rh == CD_128_8W || (rh == CD_128 && one_per_cu)) conv_launch_dma_t<2, 4, 1, 1>(a, KS, out16, s);
    else if (rh == CD_128_4BUF) conv_launch_dma_t<2, 4, 1>(a, KS, out16, s);
    else conv_launch_dma_t<2, 2, 1>(a, KS, out16, s);
}
