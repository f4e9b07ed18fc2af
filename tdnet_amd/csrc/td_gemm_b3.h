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
// one K step ahead of its MFMAs: 44 VALU instructions per 32 x 16 fragment (3 v_cvt_pk_bf16_f32, 4 shift / and, 4 v_sub_f32 per pair of
// values) against 24 MFMAs of 32 cycles -- a wave tile is 64 rows x ALL 128 columns of the workgroup tile so that a fragment is split once,
// not once per wave column.
//
// Tile 256 x 128, K step 16, four matrix waves that issue their own LDS-DMA between MFMA groups (as td_gemm_dma.h does), TWO workgroups per CU
// (253 VGPRs, 56 KB of LDS each), persistent XCD-aware tile lists as in td_gemm.h:
//   A image: [row][4 slots of 16 B] (64 bytes = the step's 16 floats), slot XOR-swizzled with (row >> 2) & 3 on the SOURCE address: every
//            lane group of a ds_read_b128 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md "LDS") covers the 16 slots of a 256-byte bank row;
//   B image: [part][k-half][128 columns][16 B], a linear copy of the packed weights; column permutation of conv_pack_weights (BN = 128, two
//            64-column groups of NT = 2), so the epilogue is td_store_acc's 16-byte path, once per column group.
// Two LDS buffers per operand; A runs one step ahead of B (it is split one step ahead): step g issues B(g + 1) and A(g + 2), reads A(g + 1)
// and B(g), and ends with vmcnt(0) + one bare barrier.  A step is written as twelve FENCED groups of four MFMAs (one product x one column
// group each); the eight pair chains of the split, the fragment reads and the wave's seven DMA pieces ride between them in a fixed order --
// left to itself the scheduler sinks the whole split behind the last MFMAs (profiles/r06b_*: hints by sched_group_barrier did not move it).
//
// Measured on MI355X (round 6, profiles/r06a_* .. r06i_*), layer 4's 512 -> 512 conv at 128 x 256 (38.7 GFLOP in the 36 GEMMs):
//   k_gemm_dma (exact fp32 MFMA)  306 us, MFMA pipes 0.88 busy, 126 TFLOP/s -- at its roof;
//   this kernel                   187 us, MFMA pipes 0.57 busy at 2.09 GHz (the chip's clock under this load), 207 TFLOP/s = 1240 TFLOP/s of
//                                 bf16 MFMA issue, what the guide's best plain-HIP bf16 GEMM reaches on random data (cdna_hip_programming.md).
// Three LDS buffers of A with the step's last four DMA pieces left in flight across the barrier (vmcnt(4) instead of 0): 0.307 vs 0.308 ms per conv, +0.3 % in the
// frame = noise (profiles/r06r_*): the wait at the end of a step is not what the pipes are waiting for.  Not kept.
// Where the other 43 % goes (in-kernel s_memtime stamps, tools/gemm_b3_trace.hip, profiles/r06s_* .. r06u_*): with two workgroups on a CU a K step of 48 MFMAs per
// wave (1536 pipe cycles) takes ~4400-5000 cycles per workgroup = issue 2100-2250 + vmcnt wait 750-1600 + barrier 800-1100.  With the A pieces zero-filled (no
// memory traffic) the conv's 36 GEMMs take 0.1925 ms, with the B pieces 0.1898, with both 0.1836, against 0.2239: the CU's LDS-DMA path carries 56 KB per ~4400 cycles,
// ~13 B/clk/CU = ~7 TB/s over the chip, and that is what the last 18 % waits for; the rest is two waves per SIMD sharing one MFMA pipe at a power-limited 2.09 GHz.
// A loaded straight from global memory into registers in fragment layout (a wave owns its 64 rows, so staging A in LDS shares nothing; two raw register sets, loop
// unrolled by two, only B on LDS-DMA: 43 % of the DMA bytes): no spills in the loop, bit-identical, and SLOWER -- the issue phase grows 2176 -> 3119 cycles (the
// register loads' vmcnt waits land inside it), 0.2299 vs 0.2224 ms per 36 GEMMs, frame 322.9 vs 325.8 at 1024x2048, 440.1 vs 444.4 at 769x1537 (profiles/r06u_*).  Not kept.
// First form of this kernel (removed; last commit with it: the one before this header's): four matrix + four LOADER waves (k_conv_dma_h3p's
// recipe), one workgroup per CU, ring of five LDS buffers: 200 us.  Leave-one-out probes of that form (profiles/r06b_* .. r06d_*): without MFMAs
// 90 us, without memory traffic 113, without the split 120, without B fragment reads 124, without epilogue stores 121, skeleton alone (barriers +
// zero-fill DMA) 21 -- MFMA time and everything else ADD UP when one workgroup owns the CU: a wave per SIMD has nobody to hide its fragment
// reads, DMA waits, barrier skew and epilogue (a quarter of a K = 512 tile).  Hence two workgroups per CU and no loader waves.
#pragma once
#include "td_gemm.h"

#include <cstring>
#include <vector>

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

// one pair of fp32 -> one dword of each part (11 VALU: 3 cvt_pk, 4 shift / and, 4 sub)
TD_DEV void td_split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = td_pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = td_pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    l = td_pk_bf16(sa, sb);
}

#ifdef TD_B3_TRACE      // tools/gemm_b3_trace.hip only: s_memtime stamps of workgroups 0..7, every wave, the first 48 K steps:
// [0] step start, [1] all MFMAs issued, [2] after the vmcnt wait, [3] after the barrier
#define TD_B3_STAMP(gs_, slot) do { if (blockIdx.x < 8 && (gs_) < 48 && lane == 0) \
    TD_B3_TRACE[(((size_t)blockIdx.x * 4 + wave) * 48 + (gs_)) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define TD_B3_DEAD(bit) ((p.act & (bit)) != 0)                      // trace builds: GemmArgs.act bit 1 / 2 = no memory traffic for the A / B pieces (zero fill)
#else
#define TD_B3_STAMP(gs_, slot) ((void)0)
#define TD_B3_DEAD(bit) false
#endif
struct GemmB3Geom {
    static constexpr int BM = 256, BN = 128;
    static constexpr int A_BYTES = BM * 64, B_BYTES = 12 * 1024, LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;
};
template <int ROLE>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_gemm_b3(GemmArgs p) {
    using G = GemmB3Geom;
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
        const bool live = ca.tile < my_tiles && !TD_B3_DEAD(1);
        td_buf_ld16_lds(a_buf, smem + abuf_i * G::A_BYTES + (wave + 4 * j) * 1024, live ? a_off[j] : TD_BUF_OOB, live ? (unsigned)ca.step * 64u : 0u);
    };
    auto end_a = [&]() {
        abuf_i ^= 1;
        if (ca.tile < my_tiles && ++ca.step == nsteps) { ca.step = 0; if (++ca.tile < my_tiles) { advance(ca.pos); enter_a(); } }
    };
    auto issue_w = [&](int jb) {
        const bool live = cw.tile < my_tiles && !TD_B3_DEAD(2);
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
    int gs = 0;                                                        // global step (trace builds)
    (void)gs;
    for (int t = 0; t < my_tiles; ++t) {
        for (int st = 0; st < nsteps; ++st) {
            TD_B3_STAMP(gs, 0);
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
            TD_B3_STAMP(gs, 1);
            TD_WAIT_VM_PIECES(0);
            TD_B3_STAMP(gs, 2);
            TD_BARRIER_RAW();
            TD_B3_STAMP(gs, 3);
            ++gs;
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

// grid_cap > 0 forces the number of workgroups (tests: several tiles per workgroup on small problems)
static inline void gemm_b3_launch(GemmArgs a, int grid_cap, hipStream_t s) {
    a.NPad = gemm_b3_npad(a.N);
    a.tiles_m = (a.M + GemmB3Geom::BM - 1) / GemmB3Geom::BM;
    a.tiles_n = a.NPad / GemmB3Geom::BN;
    const long total = (long)a.tiles_m * a.tiles_n * a.nbatch;
    long grid = grid_cap > 0 ? grid_cap : 512;                        // two resident workgroups per CU
    if (grid > total) grid = total;
    // (one tile per workgroup -- 1152 workgroups dealt out by the dispatcher as slots free up -- was measured against the 512 persistent ones:
    // 0.313 vs 0.306 ms on layer 4's 512 -> 512 conv, profiles/r06g_*)
    if (a.nbatch > 1 && !a.wshare) TD_LAUNCH((k_gemm_b3<1>), dim3((unsigned)grid), dim3(256), GemmB3Geom::LDS_BYTES, s, a);
    else if (!a.resid) TD_LAUNCH((k_gemm_b3<2>), dim3((unsigned)grid), dim3(256), GemmB3Geom::LDS_BYTES, s, a);
    else TD_LAUNCH((k_gemm_b3<0>), dim3((unsigned)grid), dim3(256), GemmB3Geom::LDS_BYTES, s, a);
}
// Which kernel form for a GEMM of `rows` x N per batch: 0 = none -- fewer than one 256 x 128 tile per CU, where the exact-fp32 kernels with
// their 64 / 128-row tiles are faster (512 -> 512 on 2048 rows: 22 us against 37; 512 -> 64: 24 against 40; profiles/r06a_*) --, else 1 = the
// matrix-only form (k_gemm_b3; 0.308 ms against 0.322 for the loader-wave form on layer 4's 512 -> 512 conv, profiles/r06e_*).  K: the reduction length (0 = unknown).
static inline int gemm_b3_pick(long rows, int nbatch, int N, int K = 0) {
    const long tiles = ((rows + 255) / 256) * (gemm_b3_npad(N) / 128) * nbatch;
    // a deep K amortises the tile's prologue and epilogue: a Bottleneck's 1024 -> 256 conv1 on 18721 rows (148 tiles, fewer than one per CU) is faster here than on the
    // fp32 GEMM -- psp101 769x1537 97.7 -> 103.2 frames/s with precision 2, td2-psp50 167.5 -> 170.9 (profiles/r06av_*)
    if (K >= 1024 && tiles >= 128 && N >= 128) return 1;
    return tiles >= 256 && N >= 128 ? 1 : 0;
}
