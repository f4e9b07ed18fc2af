// td_gemm.h -- persistent fp32-MFMA GEMM for the short-K contractions of the path:
//   out[b][m][n] = act( sum_k A[b][m][k] * W[b][k][n] + bias[n] (+ resid[m][n]) ),   b < nbatch, k < K = Cin
// i.e. every stride-1 1x1 convolution (Encoding projections transformer.py:18-24, Bottleneck 1x1s resnet.py:70-78, attention fc
// on the cached value matrix) and the 16 / 36 batched GEMMs of a Winograd conv (td_wino.h).  K is only 64..2048 here, 2..64 steps of
// 32: a workgroup that does ONE tile spends a large part of its life in the prologue (first loads exposed) and the epilogue
// (accumulator stores), which is why the conv kernel reaches 84 % of the MFMA roof on K = 4608 but 65 % on K = 512.
//
// So this kernel is persistent: the grid is one wave of resident workgroups, each walks a list of output tiles, and the
// load pipeline of td_conv.h runs ACROSS tile boundaries -- while tile i's accumulators are stored, tile i+1's
// first two K slices are already in flight / in LDS.  Same LDS images, fragment maps, weight packing (conv_pack_weights,
// KS = 1) and output-column permutation as k_conv_igemm; the A operand is a plain row-major matrix (no taps, no padding).
//
// Tile order: linear index = (batch, tile_m, tile_n) with tile_n fastest, cut into 8 contiguous ranges, one per XCD
// (workgroup w runs on XCD w % 8), so the N-tiles that share an A panel run on the same L2 at about the same time.
#pragma once
#include "td_conv.h"

struct GemmArgs {
    const float* a;       // [nbatch][M][K]
    const float* wp;      // [nbatch][K/32][8][NPad][4]  (conv_pack_weights, KS = 1)
    const float* bias;    // [N]
    const float* resid;   // [M][N] or nullptr (nbatch == 1 only)
    float* out;           // [nbatch][M][N]
    int M, N, NPad, K;
    int nbatch, act;
    int tiles_m, tiles_n; // per batch
    int MP;               // rows between consecutive batches of a and out (>= M; padded planes of the Winograd workspaces, td_wino.h)
    int wshare = 0;       // nbatch > 1 with ONE weight set and bias for all batches: a 1x1 conv over a strided set of image rows (batch = row,
                          // M = W pixels, MP = row pitch in pixels) -- the downsample conv of one row-parity chain (td_frame.h run_ds_rows); resid must be null
#ifdef TD_GEMM_TRACE      // tools/gemm_trace.hip only: per workgroup 64 x u64 -- HW_ID, XCC_ID, start, then (end of K loop, end of epilogue) per tile
    unsigned long long* trace;   // in s_memrealtime ticks (100 MHz); TD_GEMM_TRACE == 2: the end of every two-step period as well
#endif
};
#ifdef TD_GEMM_TRACE
#define TD_TRACE(slot, val) do { if (threadIdx.x == 0 && (slot) < 64) p.trace[(size_t)blockIdx.x * 64 + (slot)] = (val); } while (0)
#define TD_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define TD_TRACE(slot, val) ((void)0)
#define TD_NOW() 0ull
#endif

// ROLE names the launch for the profiler (rocprofv3 aggregates by symbol): 0 = a stride-1 1x1 convolution, 1 = the (m+2)^2
// batched GEMMs of a Winograd conv -- the frame's dominant kernel, whose roofline bench.py reports.  Same K loop; ROLE 1 has the
// plain epilogue (bias, activation and residual belong to the Winograd output transform: GemmArgs.bias / act / resid are ignored);
// ROLE 2 = a 1x1 convolution WITHOUT residual: no residual loads in the epilogue (with ROLE 0 they are issued unconditionally --
// zero-record descriptor when absent -- and, vmcnt being in order, waiting for them means waiting for the next tile's prefetch).
template <int BM, int BN, int WGM, int WGN, int ROLE>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_gemm_persistent(GemmArgs p) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int AL = BM / 32, BL = BN / 32;
    static_assert((AL == 2 || AL == 4) && (BL == 2 || BL == 4), "staging slots are spread over the 4 k-groups");
    using L = ConvLds<BM, BN>;
    TD_DYN_LDS(smem);
    float* lds = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WGN, wn = wave % WGN;
    const int a_row = tid >> 3, a_kq = tid & 7;

    // ---- this workgroup's tile list: range of XCD (bid % 8), positions q, q + G8, q + 2 G8, ... ----------------
    const int per_batch = p.tiles_m * p.tiles_n, total = per_batch * p.nbatch;
    const int NX = gridDim.x < 8 ? (int)gridDim.x : 8;               // 8 XCDs (fewer ranges only for grids smaller than that)
    const int xcd = blockIdx.x % NX, q = blockIdx.x / NX;
    const int G8 = ((int)gridDim.x + NX - 1 - xcd) / NX;             // workgroups on this XCD
    const int nq = total / NX, rem = total % NX;
    const int xbase = xcd < rem ? xcd * (nq + 1) : rem * (nq + 1) + (xcd - rem) * nq;
    const int xcount = nq + (xcd < rem ? 1 : 0);
    const int my_tiles = q < xcount ? (xcount - q + G8 - 1) / G8 : 0;
    const int nsteps = p.K >> 5;
    if (my_tiles == 0) return;
    const unsigned w_step_bytes = 8u * (unsigned)p.NPad * 16u;
    const unsigned a_bytes = (unsigned)p.M * (unsigned)p.K * 4u, w_bytes = (unsigned)nsteps * w_step_bytes;

    // ---- loader state (runs two global steps ahead of the MFMAs) -------------------------------------------------
    // Tile coordinates are advanced INCREMENTALLY (tile list = every G8-th position: add G8's (batch, tile_m, tile_n) digits with
    // carries) instead of by two integer divisions per tile (~80 dependent scalar instructions).  Measured neutral: the period that
    // enters a tile takes 1.5x either way -- its operands are cold (tools/gemm_trace.hip, warm-tile experiment in profiles/r02q_*).
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
    int l_tile = 0, l_step = 0;                                      // index into my tile list / K step inside it
    TilePos lpos = pos0;
    TdBuf a_buf, w_buf;
    unsigned a_off[AL], b_off[BL];
    auto loader_enter_tile = [&]() {
        a_buf = td_make_buf(p.a + (size_t)lpos.b * p.MP * p.K, a_bytes);
        w_buf = td_make_buf(p.wp + (p.wshare ? (size_t)0 : (size_t)lpos.b * nsteps * 8 * p.NPad * 4), w_bytes);
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int m = lpos.tm * BM + a_row + 32 * i;
            a_off[i] = m < p.M ? ((unsigned)m * (unsigned)p.K + (unsigned)a_kq * 4u) * 4u : TD_BUF_OOB;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
            b_off[i] = (unsigned)(kq * p.NPad + lpos.tn * BN + n) * 16u;
        }
    };
    auto load_tile = [&](f32x4 (&ra)[AL], f32x4 (&rb)[BL]) {
        const unsigned kb = (unsigned)l_step * 128u;                 // 32 floats per step
#pragma unroll
        for (int i = 0; i < AL; ++i) ra[i] = td_buf_ld4(a_buf, a_off[i], kb);   // the K offset rides in an SGPR: the range check looks at a_off only, so an out-of-range row stays out of range
        const unsigned wsoff = (unsigned)l_step * w_step_bytes;
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = td_buf_ld4(w_buf, b_off[i], wsoff);
        if (++l_step == nsteps) {
            l_step = 0;
            if (++l_tile < my_tiles) { advance(lpos); loader_enter_tile(); }   // past the end: stay on the last tile (never consumed)
        }
    };
    auto store_a = [&](int buf, int i, const f32x4 (&ra)[AL]) {
        td_st4(lds + buf * L::BUF_FLOATS + a_kq * L::A_STRIDE + (a_row + 32 * i) * 4, ra[i]);
    };
    auto store_b = [&](int buf, int i, const f32x4 (&rb)[BL]) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        td_st4(lds + buf * L::BUF_FLOATS + L::A_FLOATS + kq * L::B_STRIDE + n * 4, rb[i]);
    };

    f32x16 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto compute = [&](int buf, const f32x4 (&sa)[AL], const f32x4 (&sb)[BL]) {
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
#pragma unroll
            for (int i = 0; i < AL; ++i) if (i * (4 / AL) == g) store_a(buf ^ 1, i, sa);
#pragma unroll
            for (int i = 0; i < BL; ++i) if (i * (4 / BL) == g) store_b(buf ^ 1, i, sb);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = td_mfma32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j]);
            if (MT == 2 && NT == 2 && AL == 4 && BL == 4) {
                TD_SCHED_GROUP(0x008, 2); TD_SCHED_GROUP(0x200, 1); TD_SCHED_GROUP(0x008, 2); TD_SCHED_GROUP(0x200, 1);
                if (g < 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { TD_SCHED_GROUP(0x100, 1); TD_SCHED_GROUP(0x008, 3); }
                } else {
                    TD_SCHED_GROUP(0x008, 12);
                }
            }
        }
    };
    // ---- epilogue (ONE copy in the code: the K loop below is a whole number of periods) ----------------------------
    f32x4 bias_pre = {0.f, 0.f, 0.f, 0.f};                           // the current tile's bias values of this lane (td_store_acc's 16-byte path)
    const TdBuf bias_buf = td_make_buf(p.bias, (unsigned)p.N * 4u);
    const bool bias_al = (((size_t)p.bias) & 15) == 0;
    TilePos spos = pos0;                                             // the tile being multiplied
    auto fetch_bias = [&]() {                                        // at the START of a tile: the values are needed a whole K loop later,
        if (ROLE == 1 || NT != 2) return;                            // and a load issued in the epilogue would make it wait for the prefetch
        const int chan = spos.tn * BN + wn * WN + 4 * (l31 >> 1);    // of the next tile.  ROLE 1: no bias (plain epilogue).
        if (bias_al) bias_pre = td_buf_ld4(bias_buf, chan < p.N ? (unsigned)chan * 4u : TD_BUF_OOB, 0u);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_pre[e] = td_buf_ld1(bias_buf, chan + e < p.N ? (unsigned)(chan + e) * 4u : TD_BUF_OOB, 0u);
        }
    };
    auto store_tile = [&]() {
        float* outb = p.out + (size_t)spos.b * p.MP * p.N;
        td_store_acc<MT, NT, ROLE != 0, ROLE == 1>(acc, outb, p.bias, p.resid, p.M, p.N, p.act, spos.tm * BM + wm * WM, spos.tn * BN + wn * WN, lane,
                             &bias_pre);
        zero_acc();
        advance(spos);
    };

    zero_acc();
    loader_enter_tile();
#ifdef TD_GEMM_TRACE
    int tslot = 3;
    TD_TRACE(0, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_REG_HW_ID
    TD_TRACE(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));     // HW_REG_XCC_ID
    TD_TRACE(2, TD_NOW());
#endif
    // Two register sets: while global step g is multiplied, step g+1 moves registers -> LDS and step g+2 is in flight.  (A
    // third set -- three steps of lookahead -- was measured: no gain, 256 VGPRs and spills.)  K/32 is even (host check), so a
    // tile is a whole number of two-step periods and the epilogue appears once.
    f32x4 ra[AL], rb[BL], ra2[AL], rb2[BL];
    load_tile(ra, rb);                                               // global step 0
#pragma unroll
    for (int i = 0; i < AL; ++i) store_a(0, i, ra);
#pragma unroll
    for (int i = 0; i < BL; ++i) store_b(0, i, rb);
    load_tile(ra, rb);                                               // global step 1
    __syncthreads();
    for (int t = 0; t < my_tiles; ++t) {
        fetch_bias();
        int st = 0;
        do {                                                         // at least one period (nsteps >= 2): with a zero-trip path into the
            load_tile(ra2, rb2);                                     // epilogue the compiler waits there for ALL loads in flight
            compute(0, ra, rb);
            __syncthreads();
            load_tile(ra, rb);                                       // past the last tile: clamped to it, never consumed
            compute(1, ra2, rb2);
            __syncthreads();
#if defined(TD_GEMM_TRACE) && TD_GEMM_TRACE == 2
            if (st + 2 < nsteps) { TD_TRACE(tslot, TD_NOW()); ++tslot; }
#endif
        } while ((st += 2) < nsteps);
#ifdef TD_GEMM_TRACE
        TD_TRACE(tslot, TD_NOW()); ++tslot;
        store_tile();
        TD_TRACE(tslot, TD_NOW()); ++tslot;
#else
        store_tile();
#endif
    }
}

// resident workgroups per CU by LDS (65.8 KB for 128x128, 49.4 KB for the smaller tiles) and registers
static inline int gemm_blocks_per_cu(ConvTile t) { const ConvTileDims d = conv_tile_dims(t); return d.BM == 128 && d.BN == 128 ? 2 : 3; }
// the two-step period needs an even number of K steps
static inline bool gemm_supports(int K) { return K % 64 == 0; }

// Tile choice for the persistent kernel.  What matters is how many ROUNDS of the resident workgroups (256 CUs x bpc) the tile
// list takes, and how full the last round is: a CU working on j < bpc tiles finishes them faster, but not j/bpc faster (one
// workgroup alone keeps the MFMA pipes ~62 % busy, two of three ~90 %).  Measured on the F(4x4) GEMMs of the frame
// (tools/wino_tile_probe.py): equal rounds -> the three shapes tie; 256-channel layers at 769x1537: 64x128 0.093 ms vs 128x128
// 0.110 ms (792 tiles of 128x128 = 1.55 rounds of 512).
static inline ConvTile gemm_pick_tile(long rows, int nbatch, int N, bool deep) {
    if (N <= 64) return deep ? CT_128x64_DEEP : CT_128x64;
    static const struct { ConvTile t; int bm, bn, bpc; double eff; } cand[3] = {
        {CT_128x128, 128, 128, 2, 1.00}, {CT_64x128, 64, 128, 3, 0.97}, {CT_128x64, 128, 64, 3, 0.95}};
    ConvTile best = CT_128x128;
    double best_cost = 0.0;
    for (int i = 0; i < 3; ++i) {
        const long tiles = ((rows + cand[i].bm - 1) / cand[i].bm) * ((N + cand[i].bn - 1) / cand[i].bn) * nbatch;
        const long slots = 256L * cand[i].bpc, full = tiles / slots, rem = tiles % slots;
        const int j = (int)((rem + 255) / 256);
        const double u = j == cand[i].bpc ? 1.0 : j == 1 ? 0.62 : 0.9;
        const double cost = (double)cand[i].bm * cand[i].bn * (full * cand[i].bpc + (j ? j / u : 0.0)) / cand[i].eff;
        if (i == 0 || cost < best_cost) { best_cost = cost; best = cand[i].t; }
    }
    return deep ? (ConvTile)(best + 3) : best;
}

template <int BM, int BN, int WGM, int WGN>
static inline void gemm_launch_t(GemmArgs a, int bpc, int grid_cap, hipStream_t s) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = a.NPad / BN;
    const long total = (long)a.tiles_m * a.tiles_n * a.nbatch;
    long grid = grid_cap > 0 ? grid_cap : 256L * bpc;
    if (grid > total) grid = total;
    if (a.nbatch > 1 && !a.wshare) TD_LAUNCH((k_gemm_persistent<BM, BN, WGM, WGN, 1>), dim3((unsigned)grid), dim3(256), (ConvLds<BM, BN>::BYTES), s, a);
    else if (!a.resid) TD_LAUNCH((k_gemm_persistent<BM, BN, WGM, WGN, 2>), dim3((unsigned)grid), dim3(256), (ConvLds<BM, BN>::BYTES), s, a);
    else TD_LAUNCH((k_gemm_persistent<BM, BN, WGM, WGN, 0>), dim3((unsigned)grid), dim3(256), (ConvLds<BM, BN>::BYTES), s, a);
}
// grid_cap > 0 forces the number of workgroups (tests: several tiles per workgroup on small problems)
static inline void gemm_launch(const GemmArgs& a, ConvTile tile, int grid_cap, hipStream_t s) {
    const ConvTileDims d = conv_tile_dims(tile);
    const int bpc = gemm_blocks_per_cu(tile);
    if (d.BM == 128 && d.BN == 128) gemm_launch_t<128, 128, 2, 2>(a, bpc, grid_cap, s);
    else if (d.BM == 64) gemm_launch_t<64, 128, 2, 2>(a, bpc, grid_cap, s);
    else gemm_launch_t<128, 64, 4, 1>(a, bpc, grid_cap, s);
}
