// td_conv.h -- NHWC implicit-GEMM convolution on the gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Covers every conv on the TDNet hot path: 3x3 with stride 1/2 and dilation
// 1..16 (resnet.py:32-37), 1x1 with stride 1/2/4 (downsample resnet.py:172-177, Encoding transformer.py:18-26,36),
// and the 7x7-s2 stem (resnet.py:132-133) through the STEM variant.  BN is folded into weight/bias on the host, so the
// epilogue is  out = act(acc + bias[n] (+ residual[m][n])).
//
//   GEMM view:  M = Ho*Wo output pixels, N = Cout, K = taps * Cin.   D[m][n] = sum_k A[m][k] * B[k][n]
//   A[m][k] is gathered on the fly from the NHWC input (zero outside the image), B is the pre-packed weight.
//
// Data movement per block (256 threads = 4 waves, BM x BN output tile, BK = 32 reduction slice per step):
//   global -> registers (float4, issued one step ahead) -> LDS (double buffered, one barrier per step) -> MFMA operands.
//   LDS images are [kq = 8 groups of 4 consecutive k][row][4 floats]: an MFMA lane (row = lane&31, half = lane>>5) reads
//   ONE ds_read_b128 per operand per 4 MFMAs, conflict free (consecutive rows = consecutive 16-B slots).  Inside a group
//   of 8 k the two lane-halves take k = 4*half + s, s = 0..3: a fixed permutation of the reduction order, applied to A
//   and B alike, so the sum is unchanged.
//   The K loop runs channel-chunk outer / tap inner so the 9 shifted re-reads of an input slab hit L2 back to back.
//   Output columns are permuted inside a wave (lane j owns NT consecutive channels j*NT..j*NT+NT-1) by the weight
//   packer, so the epilogue stores NT*4 contiguous bytes per lane instead of 4.
#pragma once
#include "td_device.h"

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
    int nsteps;           // (Cin/32)*KS*KS, STEM: 7
    int act;              // 0 none, 1 ReLU, 2 LeakyReLU(0.01)
    int tiles_n;          // CoutPad / BN
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

template <int BM, int BN, int WGM, int WGN, int KS, bool STEM>
TD_KERNEL void TD_LAUNCH_BOUNDS(256, 2) k_conv_igemm(ConvArgs p) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int AL = BM / 32, BL = BN / 32;          // float4 slots per thread per step
    using L = ConvLds<BM, BN>;
    constexpr int NTAPS = STEM ? 1 : KS * KS;
    TD_DYN_LDS(smem);
    float* lds = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WGN, wn = wave % WGN;

    const int lin = td_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lin / p.tiles_n, tile_n = lin - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread gather geometry: slot i -> row (tid>>3) + 32 i, k-group tid&7
    const int a_row = tid >> 3, a_kq = tid & 7;
    int a_by[AL], a_bx[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + a_row + 32 * i;
        const int oy = m / p.Wo, ox = m - oy * p.Wo;
        a_by[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 28);
        a_bx[i] = ox * p.stride - p.pad;
    }
    const float* wbase = p.wp + (size_t)n0 * 4;

    f32x4 ra[AL], rb[BL];
    auto load_tile = [&](int step, int chunk, int tap) {
        int dy, dx, coff;
        bool tap_ok = true;
        if (STEM) {
            const int t = step * 8 + a_kq;             // one 4-channel pixel per k-group: 8 taps per step
            const int ky = t / 7;
            dy = ky; dx = t - ky * 7; coff = 0; tap_ok = t < 49;
        } else {
            const int ky = tap / KS;
            dy = ky * p.dil; dx = (tap - ky * KS) * p.dil; coff = chunk * 32 + a_kq * 4;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int iy = a_by[i] + dy, ix = a_bx[i] + dx;
            const bool ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = td_ld4(p.in + ((size_t)(iy * p.W + ix) * p.Cin + coff));
            ra[i] = v;
        }
        const float* wsrc = wbase + (size_t)step * 8 * p.CoutPad * 4;
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
            rb[i] = td_ld4(wsrc + ((size_t)kq * p.CoutPad + n) * 4);
        }
    };
    auto store_tile = [&](int buf) {
        float* As = lds + buf * L::BUF_FLOATS;
        float* Bs = As + L::A_FLOATS;
#pragma unroll
        for (int i = 0; i < AL; ++i) td_st4(As + a_kq * L::A_STRIDE + (a_row + 32 * i) * 4, ra[i]);
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i, kq = idx / BN, n = idx % BN;
            td_st4(Bs + kq * L::B_STRIDE + n * 4, rb[i]);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int chunk = 0, tap = 0;
    load_tile(0, 0, 0);
    store_tile(0);
    __syncthreads();
    for (int step = 0; step < p.nsteps; ++step) {
        const bool more = step + 1 < p.nsteps;
        if (more) {
            if (++tap == NTAPS) { tap = 0; ++chunk; }
            load_tile(step + 1, chunk, tap);            // global loads in flight while the MFMAs below run
        }
        const float* As = lds + (step & 1) * L::BUF_FLOATS + (wm * WM + l31) * 4;
        const float* Bs = lds + (step & 1) * L::BUF_FLOATS + L::A_FLOATS + (wn * WN + l31) * 4;
        // operand fragments are fetched one k-group ahead of the MFMAs that consume them (register double buffer)
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
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = td_mfma32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j]);
            // pin the interleave: the next group's LDS reads are spread between this group's MFMAs instead of being
            // sunk to their first use (where every group would start with an exposed LDS round trip)
            if (g < 3) {
#pragma unroll
                for (int r = 0; r < MT + NT; ++r) {
                    TD_SCHED_GROUP(0x100, 1);
                    TD_SCHED_GROUP(0x008, (4 * MT * NT) / (MT + NT));
                }
            } else {
                TD_SCHED_GROUP(0x008, 4 * MT * NT);
            }
        }
        if (more) store_tile((step + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns channels nb .. nb+NT-1 of row m (see weight packer) ----
    const int nb = n0 + wn * WN + l31 * NT;
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = (nb + j < p.Cout) ? p.bias[nb + j] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= p.M) continue;
            const size_t o = (size_t)m * p.Cout + nb;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (nb + j >= p.Cout) continue;
                float v = acc[i][j][r] + bv[j];
                if (p.resid) v += p.resid[o + j];
                if (p.act == 1) v = v > 0.f ? v : 0.f;
                else if (p.act == 2) v = v > 0.f ? v : 0.01f * v;
                p.out[o + j] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side: tile configuration, weight packing, launch
// ------------------------------------------------------------------------------------------------------------------
enum ConvTile { CT_128x128 = 0, CT_64x128 = 1, CT_128x64 = 2 };

struct ConvTileDims { int BM, BN, WGM, WGN; };
static inline ConvTileDims conv_tile_dims(ConvTile t) {
    switch (t) {
        case CT_128x128: return {128, 128, 2, 2};
        case CT_64x128: return {64, 128, 2, 2};
        default: return {128, 64, 4, 1};
    }
}

// Choose the tile: 128-wide N when Cout allows, and the smaller M tile when 128x128 would leave CUs idle.
static inline ConvTile conv_pick_tile(int M, int Cout) {
    if (Cout <= 64) return CT_128x64;
    const int tn = (Cout + 127) / 128;
    const long blocks = (long)((M + 127) / 128) * tn;
    return blocks >= 512 ? CT_128x128 : CT_64x128;
}

static inline int conv_cout_pad(int Cout, ConvTile t) {
    const int BN = conv_tile_dims(t).BN;
    return (Cout + BN - 1) / BN * BN;
}
static inline int conv_nsteps(int Cin, int KS, bool stem) { return stem ? 7 : (Cin / 32) * KS * KS; }

// Pack BN-folded OIHW weights into the LDS image order of `tile`: [step][kq][slot][4], where packed column `slot`
// holds output channel  tile_n*BN + wn*WN + j*NT + nt   for  slot = tile_n*BN + wn*WN + nt*32 + j.
static inline void conv_pack_weights(const float* w, int Cout, int Cin, int KS, bool stem, ConvTile tile, float* dst) {
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
                        if (stem) {
                            const int t = step * 8 + kq;
                            if (t < 49 && e < 3) v = w[((size_t)n * 3 + e) * 49 + t];
                        } else {
                            const int chunk = step / ntaps, tap = step % ntaps, ci = chunk * 32 + kq * 4 + e;
                            v = w[((size_t)n * Cin + ci) * ntaps + tap];
                        }
                    }
                    o[e] = v;
                }
            }
}

template <int BM, int BN, int WGM, int WGN>
static inline void conv_launch_t(const ConvArgs& a, int KS, bool stem, hipStream_t s) {
    const int grid = ((a.M + BM - 1) / BM) * a.tiles_n;
    const int lds = ConvLds<BM, BN>::BYTES;
    if (stem) TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 7, true>), dim3(grid), dim3(256), lds, s, a);
    else if (KS == 3) TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 3, false>), dim3(grid), dim3(256), lds, s, a);
    else TD_LAUNCH((k_conv_igemm<BM, BN, WGM, WGN, 1, false>), dim3(grid), dim3(256), lds, s, a);
}

static inline void conv_launch(ConvArgs a, ConvTile tile, int KS, bool stem, hipStream_t s) {
    const ConvTileDims d = conv_tile_dims(tile);
    a.tiles_n = a.CoutPad / d.BN;
    switch (tile) {
        case CT_128x128: conv_launch_t<128, 128, 2, 2>(a, KS, stem, s); break;
        case CT_64x128: conv_launch_t<64, 128, 2, 2>(a, KS, stem, s); break;
        default: conv_launch_t<128, 64, 4, 1>(a, KS, stem, s); break;
    }
}
