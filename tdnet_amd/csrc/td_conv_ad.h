// td_conv_ad.h -- implicit-GEMM convolution for the Cout <= 64 layers (ResNet layer1, the 7x7 / 3x3 stems) with the A operand
// read STRAIGHT from global memory in MFMA fragment layout (tdnet_opts.fusion bit 32).
//
// Why: with a 128 x 64 block (four waves of 32 rows x 64 channels, MT = 1, NT = 2) k_conv_igemm moves, per MFMA, 1.5x the LDS bytes
// of the 128 x 128 block (three 16-byte fragment reads per 8 MFMAs instead of four per 16, plus the staging writes): at three
// resident workgroups that is ~80 % of the CU's LDS bandwidth, and the kernel sits at 95 TFLOP/s where the 128 x 128 kernel reaches
// 130.  Here a lane's A fragment for k-group pair g is 16 contiguous bytes of ITS OWN output pixel's (tap-shifted) NHWC row --
// channels 8g + 4 half .. + 3 -- so it can be loaded directly as the MFMA operand: no LDS write, no LDS read, no redundancy (with
// WGN = 1 no two waves share a row).  Padded taps use the buffer bounds check (zeros).  Only the 64-column weight tile goes through
// LDS (16 KB for both buffers), in the image and packing of td_conv.h (tile CT_128x64).  Same fp32 MFMA, same reduction order
// inside a K step, same epilogue: results are bit-identical to k_conv_igemm<128, 64, 4, 1, ...>.
#pragma once
#include "td_conv.h"

// STEM: 0 = an NHWC map of Cin % 32 == 0 channels; 1 = the stems on the NHWC4 image (one 4-channel pixel = one tap per k-group, 8 taps per K
// step); 2 (round 5) = the 7x7 stem on the PACKED-ROW image: [H + 7][Wp][3] floats with a zero border (stem_rows_wp below), so the 7 taps x 3
// channels of a kernel row are 21 CONTIGUOUS floats -- a K step is one kernel row: six 16-byte k-groups (21 products, three columns and
// two groups of zero weights), 24 MFMAs instead of 32; 7 steps, K = 168 instead of 224 for the same 147 products.  The loads are 4-byte
// aligned 16-byte buffer loads; nothing is masked (the border is in the image, a row past M reads out of range = zeros).
template <int KS, int STEM>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 3) k_conv_adirect(ConvArgs p) {
    constexpr int BM = 128, BN = 64, NT = 2;
    constexpr int NTAPS = STEM ? 1 : KS * KS;
    constexpr int NG = STEM == 2 ? 3 : 4;                            // k-group pairs per K step
    constexpr int B_STRIDE = BN * 4, BUF_FLOATS = 8 * B_STRIDE;     // weights only: [kq][64 slots][4 floats]
    TD_DYN_LDS(smem);
    float* lds = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // this lane's output pixel = row (wave, l31) of the block
    const int m = m0 + wave * 32 + l31;
    const int oy = m / p.Wo, ox = m - oy * p.Wo;
    const int a_by = (m < p.M) ? oy * p.stride - p.pad : -(1 << 28);
    const int a_bx = ox * p.stride - p.pad + (STEM == 2 ? 1 : 0);   // packed-row image: 4 border pixels on the left (16-byte aligned rows), 3 needed
    const unsigned a_off = (((unsigned)a_by * (unsigned)p.W + (unsigned)a_bx) * (unsigned)p.Cin + (STEM ? 0u : (unsigned)half * 4u)) * 4u;
    const TdBuf in_buf = td_make_buf(p.in, (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cin * 4u);
    const TdBuf w_buf = td_make_buf(p.wp, (unsigned)p.nsteps * 8u * (unsigned)p.CoutPad * 16u);
    unsigned b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        b_off[i] = (unsigned)(kq * p.CoutPad + n0 + n) * 16u;
    }
    const unsigned w_step_bytes = 8u * (unsigned)p.CoutPad * 16u;

    // A fragments of step la_step: register g = k-groups (2g, 2g+1), this lane's half picks one; advanced after every call
    int la_step = 0, la_chunk = 0, la_tap = 0;
    auto load_a = [&](f32x4 (&ra)[4]) {
        const bool live = la_step < p.nsteps;                       // wave-uniform: past the last step nothing is consumed -> read zeros
        if (STEM == 2) {
            const bool ok = live && m < p.M;
#pragma unroll
            for (int g = 0; g < NG; ++g)                             // kernel row la_step: floats 4 (2 g + half) .. + 3 of its 24
                ra[g] = td_buf_ld4(in_buf, ok ? a_off + (unsigned)(la_step * p.W * 3 + (2 * g + half) * 4) * 4u : TD_BUF_OOB, 0u);
        } else if (STEM) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int t = la_step * 8 + 2 * g + half;           // one 4-channel pixel per k-group: 8 taps per step
                const int ky = t / KS, dy = ky, dx = t - ky * KS;
                const int iy = a_by + dy, ix = a_bx + dx;
                const bool ok = live && t < KS * KS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                ra[g] = td_buf_ld4(in_buf, ok ? a_off + (unsigned)((dy * p.W + dx) * 4) * 4u : TD_BUF_OOB, 0u);
            }
        } else {
            const int ky = la_tap / KS;
            const int dy = ky * p.dil, dx = (la_tap - ky * KS) * p.dil;
            const int iy = a_by + dy, ix = a_bx + dx;
            const bool ok = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = ok ? a_off + (unsigned)((dy * p.W + dx) * p.Cin + la_chunk * 32) * 4u : TD_BUF_OOB;
#pragma unroll
            for (int g = 0; g < 4; ++g) ra[g] = td_buf_ld4(in_buf, off, (unsigned)g * 32u);   // channels 8g + 4 half .. + 3 of the chunk
        }
        ++la_step;
        if (++la_tap == NTAPS) { la_tap = 0; ++la_chunk; }
    };
    int lb_step = 0;
    auto load_b = [&](f32x4 (&rb)[2]) {
        const unsigned wsoff = (unsigned)(lb_step < p.nsteps ? lb_step : p.nsteps - 1) * w_step_bytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = td_buf_ld4(w_buf, b_off[i], wsoff);
        ++lb_step;
    };
    auto store_b = [&](int buf, int i, const f32x4 (&rb)[2]) {
        const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
        td_st4(lds + buf * BUF_FLOATS + kq * B_STRIDE + n * 4, rb[i]);
    };

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // one K step on weight buffer `buf` with A fragments `af`; the staged weights `sb` (tile s+1) go to buffer buf^1 in between
    auto compute = [&](int buf, const f32x4 (&af)[4], const f32x4 (&sb)[2]) {
        const float* Bs = lds + buf * BUF_FLOATS + l31 * 4;
        f32x4 bf[2][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[0][j] = td_ld4(Bs + half * B_STRIDE + j * 128);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g < NG - 1) {
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[(g + 1) & 1][j] = td_ld4(Bs + (2 * g + 2 + half) * B_STRIDE + j * 128);
            }
            if (g == 0) store_b(buf ^ 1, 0, sb);
            if (g == 2) store_b(buf ^ 1, 1, sb);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[0][j] = td_mfma32(af[g][s], bf[g & 1][j][s], acc[0][j]);
        }
    };

    // A one step ahead in registers (two sets), weights two steps ahead (register set -> LDS buffer), one barrier per step
    f32x4 a0[4], a1[4], rb[2], rb2[2];
    load_a(a0);                                                     // step 0
    load_b(rb);                                                     // weights of step 0
#pragma unroll
    for (int i = 0; i < 2; ++i) store_b(0, i, rb);
    load_b(rb);                                                     // weights of step 1
    __syncthreads();
    int step = 0;                                                   // whole periods, then the odd last step (td_conv.h)
    for (; step + 1 < p.nsteps; step += 2) {
        load_a(a1);                                                 // A of step+1, weights of step+2
        load_b(rb2);
        compute(0, a0, rb);
        __syncthreads();
        load_a(a0);                                                 // A of step+2, weights of step+3
        load_b(rb);
        compute(1, a1, rb2);
        __syncthreads();
    }
    if (step < p.nsteps) compute(0, a0, rb);
    td_store_acc<1, NT>(acc, p.out, p.bias, p.resid, p.M, p.Cout, p.act, m0 + wave * 32, n0, lane);
}

// supported: the 128 x 64 tiles (weights packed for CT_128x64 / CT_128x64_DEEP: same packing), no batching
static inline bool conv_adirect_supports(ConvTile tile, int nbatch) { return (tile == CT_128x64 || tile == CT_128x64_DEEP) && nbatch <= 1; }

// stem: 0 no, 1 the NHWC4 image, 2 the packed-row image (7x7 only; the caller passes the padded image's H, W, Cin = 3, pad = 0)
static inline void conv_launch_adirect(ConvArgs a, int KS, int stem, hipStream_t s) {
    a.tiles_n = a.CoutPad / 64;
    const int grid = ((a.M + 127) / 128) * a.tiles_n;
    const int lds = 2 * 8 * 64 * 4 * 4;
    if (stem == 2 && KS == 7) TD_LAUNCH((k_conv_adirect<7, 2>), dim3(grid), dim3(256), lds, s, a);
    else if (stem && KS == 7) TD_LAUNCH((k_conv_adirect<7, 1>), dim3(grid), dim3(256), lds, s, a);
    else if (stem) TD_LAUNCH((k_conv_adirect<3, 1>), dim3(grid), dim3(256), lds, s, a);
    else if (KS == 3) TD_LAUNCH((k_conv_adirect<3, 0>), dim3(grid), dim3(256), lds, s, a);
    else TD_LAUNCH((k_conv_adirect<1, 0>), dim3(grid), dim3(256), lds, s, a);
}
// the packed-row image of an H x W frame: [H + 7][Wp][3] floats -- 3 border rows on top, FOUR border pixels on the left (pixel x sits at
// column x + 4, so that 4 consecutive pixels = 12 floats start on a 16-byte boundary and the layout kernel writes whole 16-byte vectors),
// and on the right / bottom what the last 24-float row read reaches; Wp a multiple of 4 (every row starts on a 16-byte boundary)
static inline int stem_rows_hp(int H) { return H + 7; }
static inline int stem_rows_wp(int W) { return (W + 9 + 3) / 4 * 4; }
