// td_conv_ad_b3.h -- tdnet_opts.precision = 2 for the Cout <= 64 convs (ResNet layer1: four 64 -> 64 3x3 convs on the 1/4-resolution map, 12 % of a
// td4-psp18 frame once the deeper layers run on the split GEMM): k_conv_adirect's structure (td_conv_ad.h: the A operand straight from global memory in
// MFMA fragment layout, only the 64-column weight tile through LDS) on the bf16 MFMA with both operands as three bf16 parts (td_gemm_b3.h: six products
// per fp32 product, fp32 accumulation, every product carried to ~2^-26).
//
// A K step is 32 channels of one tap = two 32x32x16 MFMA k-blocks.  A lane's fragment of block b is channels  32 chunk + 16 b + 8 half .. + 7  of ITS OWN
// output pixel's tap-shifted NHWC row: two 16-byte buffer loads, split in registers (td_split3, 44 VALU per block) one step after they were issued.
// Weights are split on the host (conv_pack_weights_adb3): per step [part 3][k block 2][k half 2][CoutPad][8 bf16], 12 KB per 64-column tile, brought by LDS-DMA
// two steps ahead into one of three LDS buffers (one bare barrier per step); a B fragment is one ds_read_b128.  Per wave and step: 24 MFMAs of 32 cycles against the
// fp32 kernel's 32 of 64.  Column permutation and epilogue are k_conv_adirect's (td_store_acc<1, 2>).  Not bit-identical to the fp32 kernel (the matrix
// core's own summation order inside an instruction): held to the same gates against fp64 (tests/test_gpu_b3.py).
#pragma once
#include "td_conv_ad.h"
#include "td_gemm_b3.h"

#ifndef TD_ADB3_SKIP      // tools/adb3_skip_probe.hip only: leave-one-out timing builds (1 no split, 2 no A loads, 4 no B fragment reads, 8 no weight DMA, 16 no MFMAs)
#define TD_ADB3_SKIP 0
#endif
#ifndef TD_ADB3_OCC
#define TD_ADB3_OCC 4
#endif
// (A 128-column tile -- NT = 4, a wave holds 32 rows x 128 channels, A loaded and split once for a conv of 65 .. 128 output channels instead of once per 64-column
// tile; 170 VGPRs, two workgroups per CU -- was built and measured neutral: td4-psp18 352.4 vs 353.3 frames/s, td2-psp50 169.0 vs 168.8, psp101 102.0 vs 101.7
// (profiles/r06ax_*).  Removed.)
// STEM = 2: the 7x7 stride-2 stem on the packed-row image (td_conv_ad.h): a K step is one kernel row, 32 consecutive floats of the image row from the output
// pixel's first tap -- 21 products (7 taps x 3 channels) and 11 columns of zero weights (the fp32 kernel: 24 floats, 3 zero columns).
template <int KS, int STEM>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, TD_ADB3_OCC) k_conv_adirect_b3(ConvArgs p) {
    constexpr int BM = 128, BN = 64, NT = 2;
    constexpr int NTAPS = STEM ? 1 : KS * KS;
    constexpr int BUF_BYTES = 12 * BN * 16;                           // weights only: [part][k block][k half][64 slots][16 B]; THREE buffers
    TD_DYN_LDS(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int m = m0 + wave * 32 + l31;                              // this lane's output pixel = row (wave, l31) of the block
    const int oy = m / p.Wo, ox = m - oy * p.Wo;
    const int a_by = (m < p.M) ? oy * p.stride - p.pad : -(1 << 28);
    const int a_bx = ox * p.stride - p.pad + (STEM == 2 ? 1 : 0);   // packed-row image: 4 border pixels on the left (16-byte aligned rows), 3 needed
    const unsigned a_off = (((unsigned)a_by * (unsigned)p.W + (unsigned)a_bx) * (unsigned)p.Cin + (unsigned)half * 8u) * 4u;
    const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 4u);
    const unsigned w_step_bytes = 12u * (unsigned)p.CoutPad * 16u;
    const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * w_step_bytes);
    unsigned b_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) b_off[i] = (unsigned)((wave + 4 * i) * p.CoutPad + n0 + lane) * 16u;   // LDS-DMA piece wave + 4 i = image row (part, k block, k half)

    // raw A fragments of step la_step: raw[2 b + e] = channels 16 b + 8 half + 4 e .. + 3 of the step's 32; advanced after every call
    struct ASet { f32x4 raw[4]; u32x4 h0, m0, l0; };                  // a step's raw fragments and the three parts of its k block 0
    int la_step = 0, la_chunk = 0, la_tap = 0;
    unsigned la_off = 0;                                              // this lane's byte offset of step la_step's 32 floats (or out of range)
    auto addr_a = [&]() {
        const bool live = la_step < p.nsteps;                       // wave-uniform: past the last step nothing is consumed -> read zeros
        if (STEM == 2) {                                            // kernel row la_step: floats 16 b + 8 half + 4 e .. + 3 of its 32 (4-byte aligned 16-byte loads)
            la_off = (live && m < p.M) ? a_off + (unsigned)(la_step * p.W * 3) * 4u : TD_BUF_OOB;
            return;
        }
        const int ky = la_tap / KS;
        const int dy = ky * p.dil, dx = (la_tap - ky * KS) * p.dil;
        const int iy = a_by + dy, ix = a_bx + dx;
        const bool ok = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        la_off = ok ? a_off + (unsigned)((dy * p.W + dx) * p.Cin + la_chunk * 32) * 4u : TD_BUF_OOB;
    };
    auto piece_a = [&](ASet& a, int q) {                             // raw[q], q = 2 b + e
        if (!(TD_ADB3_SKIP & 2) || la_step == 0) a.raw[q] = td_buf_ld4(in_buf, la_off, (unsigned)((q >> 1) * 64 + (q & 1) * 16));
        else TD_PIN(a.raw[q]);
    };
    auto end_a = [&]() {
        ++la_step;
        if (!STEM && ++la_tap == NTAPS) { la_tap = 0; ++la_chunk; }
    };
    auto load_a = [&](ASet& a) {
        addr_a();
#pragma unroll
        for (int q = 0; q < 4; ++q) piece_a(a, q);
        end_a();
    };
    int lb_step = 0, lb_buf = 0;                                      // the weights of step lb_step go to LDS buffer lb_buf = lb_step % 3
    auto piece_w = [&](int i) {
        const bool live = lb_step < p.nsteps;
        if (!(TD_ADB3_SKIP & 8) || lb_step < 3)
            td_buf_ld16_lds(w_buf, smem + lb_buf * BUF_BYTES + (wave + 4 * i) * 1024, live ? b_off[i] : TD_BUF_OOB, live ? (unsigned)lb_step * w_step_bytes : 0u);
    };
    auto end_w = [&]() {
        ++lb_step;
        lb_buf = lb_buf == 2 ? 0 : lb_buf + 1;
    };
    auto issue_w = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) piece_w(i);
        end_w();
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto mm = [&](const u32x4& a, const u32x4 (&b)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (TD_ADB3_SKIP & 16) { unsigned a0 = a[0], b0 = b[j][0]; asm volatile("" :: "v"(a0), "v"(b0)); }
            else acc[j] = td_mfma32_bf16(a, b[j], acc[j]);
        }
    };
    const unsigned frag_off = (unsigned)(l31 * 16 + half * (BN * 16));
    u32x4 pf[NT];                                                     // B fragment (high part, k block 0) of the coming step, read at the end of the one before
    int cb = 0;                                                       // LDS buffer of the current step
    auto frag = [&](int buf, int part, int b, u32x4 (&f)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if ((TD_ADB3_SKIP & 4) && la_step > 1) f[j] = pf[j];
            else f[j] = *reinterpret_cast<const u32x4*>(smem + buf * BUF_BYTES + frag_off + ((part * 2 + b) * 2) * (BN * 16) + j * 512);
        }
    };
    auto pair = [&](const f32x4& x0, const f32x4& x1, int i, u32x4& h, u32x4& mdl, u32x4& l) {   // values 2 i, 2 i + 1 of a lane's 8 -> dword i of each part
        const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
        unsigned h_, m_, l_;
        if (TD_ADB3_SKIP & 1) { h_ = __builtin_bit_cast(unsigned, a); m_ = __builtin_bit_cast(unsigned, b); l_ = h_; }
        else td_split3_pair(a, b, h_, m_, l_);
        h[i] = h_; mdl[i] = m_; l[i] = l_;
    };                                                     // B fragment (high part, k block 0) of the coming step, read at the end of the one before

    // One K step = twelve fenced groups of two MFMAs (one product x the two 32-column halves).  Between them: the loads of the NEXT step's raw A and the LDS-DMA of
    // the weights two steps ahead (group 0), the split of this step's k block 1 (groups 0-3) and of the next step's k block 0 (groups 6-9), the B fragments one
    // to three groups ahead of their MFMAs.  One bare barrier at the top: it publishes the weights issued during the previous step (each wave waits for its own
    // pieces first) to the step AFTER this one's -- three buffers, so nothing written after it is read before the next barrier and nothing read after it is
    // overwritten before the next barrier.  (The step's seven vector-memory requests one per group instead of all in group 0: 59.5 vs 60.0 us, profiles/r06ad_*.)
    auto kstep = [&](ASet& C, ASet& N) {
        TD_WAIT_VM_PIECES(0);
        TD_BARRIER_RAW();
        const int nb = cb == 2 ? 0 : cb + 1;
        u32x4 bh[NT] = {pf[0], pf[1]}, bm[NT], bl[NT], ch[NT], cm[NT], cl[NT], h1, m1, l1;
        TD_SCHED_FENCE();
        mm(C.h0, bh); load_a(N); issue_w(); frag(cb, 1, 0, bm); pair(C.raw[2], C.raw[3], 0, h1, m1, l1);
        TD_SCHED_FENCE();
        mm(C.m0, bh); pair(C.raw[2], C.raw[3], 1, h1, m1, l1);
        TD_SCHED_FENCE();
        mm(C.l0, bh); frag(cb, 2, 0, bl); pair(C.raw[2], C.raw[3], 2, h1, m1, l1);
        TD_SCHED_FENCE();
        mm(C.h0, bm); pair(C.raw[2], C.raw[3], 3, h1, m1, l1);
        TD_SCHED_FENCE();
        mm(C.m0, bm); frag(cb, 0, 1, ch);
        TD_SCHED_FENCE();
        mm(C.h0, bl); frag(cb, 1, 1, cm);
        TD_SCHED_FENCE();
        mm(h1, ch); pair(N.raw[0], N.raw[1], 0, N.h0, N.m0, N.l0);
        TD_SCHED_FENCE();
        mm(m1, ch); pair(N.raw[0], N.raw[1], 1, N.h0, N.m0, N.l0);
        TD_SCHED_FENCE();
        mm(l1, ch); frag(cb, 2, 1, cl); pair(N.raw[0], N.raw[1], 2, N.h0, N.m0, N.l0);
        TD_SCHED_FENCE();
        mm(h1, cm); pair(N.raw[0], N.raw[1], 3, N.h0, N.m0, N.l0);
        TD_SCHED_FENCE();
        mm(m1, cm); frag(nb, 0, 0, pf);
        TD_SCHED_FENCE();
        mm(h1, cl);
        TD_SCHED_FENCE();
        cb = nb;
    };

    ASet X, Y;
    load_a(X);                                                      // step 0
    issue_w(); issue_w();                                           // weights of steps 0 and 1
    TD_WAIT_VM_PIECES(0);
    TD_BARRIER_RAW();
#pragma unroll
    for (int i = 0; i < 4; ++i) pair(X.raw[0], X.raw[1], i, X.h0, X.m0, X.l0);
    frag(0, 0, 0, pf);
    int step = 0;                                                   // whole periods, then the odd last step
    for (; step + 1 < p.nsteps; step += 2) { kstep(X, Y); kstep(Y, X); }
    if (step < p.nsteps) kstep(X, Y);
    f32x16 accs[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) accs[0][j] = acc[j];
    td_store_acc<1, NT>(accs, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wave * 32, n0, lane);
}

static inline bool conv_adirect_b3_supports(ConvTile tile, int Cin, int KS, bool stem) {
    return conv_adirect_supports(tile, 1) && !stem && Cin % 32 == 0 && (KS == 1 || KS == 3);
}
// stem = 2: the 7x7 stem on the packed-row image, KS steps
static inline size_t conv_adb3_packed_bytes(int Cout, int Cin, int KS, int stem = 0) { return (size_t)conv_nsteps(Cin, KS, stem) * 12 * ((Cout + 63) / 64 * 64) * 16; }
// BN-folded OIHW weights -> [step = chunk * KS^2 + tap][part][k block][k half][CoutPad][8 bf16]; packed column `slot` holds output channel
// tn * 64 + j * 2 + nt  for  slot = tn * 64 + nt * 32 + j  (conv_pack_weights on the 128 x 64 tile)
// stem = 2 (w is [Cout][3][KS][KS]): step = kernel row ky, k = 16 b + 8 kh + e = 3 kx + c for k < 3 KS, zero weights above
static inline void conv_pack_weights_adb3(const float* w, int Cout, int Cin, int KS, unsigned short* dst, int stem = 0) {
    const int CoutPad = (Cout + 63) / 64 * 64, ntaps = KS * KS, nsteps = conv_nsteps(Cin, KS, stem);
    for (int step = 0; step < nsteps; ++step) {
        const int chunk = step / ntaps, tap = step % ntaps;
        for (int slot = 0; slot < CoutPad; ++slot) {
            const int tn = slot / 64, w2 = slot % 64, nt = w2 / 32, j = w2 % 32;
            const int n = tn * 64 + j * 2 + nt;
            for (int b = 0; b < 2; ++b)
                for (int kh = 0; kh < 2; ++kh)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = chunk * 32 + b * 16 + kh * 8 + e, k = b * 16 + kh * 8 + e;
                        const float x = n >= Cout ? 0.f : stem != 2 ? w[((size_t)n * Cin + ci) * ntaps + tap]
                                      : k < 3 * KS ? w[((size_t)n * 3 + k % 3) * ntaps + step * KS + k / 3] : 0.f;
                        const unsigned short h = gemm_b3_bf16(x);
                        const float r = x - gemm_b3_widen(h);
                        const unsigned short mm = gemm_b3_bf16(r);
                        const unsigned short l = gemm_b3_bf16(r - gemm_b3_widen(mm));
                        const unsigned short parts[3] = {h, mm, l};
                        for (int part = 0; part < 3; ++part)
                            dst[((((size_t)step * 3 + part) * 2 + b) * 2 + kh) * CoutPad * 8 + (size_t)slot * 8 + e] = parts[part];
                    }
        }
    }
}
// stem: 0 an NHWC map, 2 the packed-row image of the 7x7 stem (the caller passes the padded image's H, W, Cin = 3, pad = 0, as for conv_launch_adirect)
static inline void conv_launch_adirect_b3(ConvArgs a, int KS, int stem, hipStream_t s) {
    a.tiles_n = a.CoutPad / 64;
    const int grid = ((a.M + 127) / 128) * a.tiles_n;
    const int lds = 3 * 12 * 64 * 16;
    if (stem == 2) TD_LAUNCH((k_conv_adirect_b3<7, 2>), dim3(grid), dim3(256), lds, s, a);
    else if (KS == 3) TD_LAUNCH((k_conv_adirect_b3<3, 0>), dim3(grid), dim3(256), lds, s, a);
    else TD_LAUNCH((k_conv_adirect_b3<1, 0>), dim3(grid), dim3(256), lds, s, a);
}
