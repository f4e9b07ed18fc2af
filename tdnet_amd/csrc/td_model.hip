// td_model.hip -- host side of libtdnet_hip.so: handle, strict weight loading + BN folding + packing, the per-frame
// forward (kernel sequence of SURVEY.md §8a rows A1-A13), the K/Q/V FIFO, and the C ABI of include/tdnet.h.
//
// Reference behaviour mirrored here (paths relative to /root/reference/Testing/model/pspnet):
//   forward / path dispatch ........ td4_psp18.py:216-229, td2_psp50.py:146-155
//   per-path graph ................. td4_psp18.py:137-212, td2_psp50.py:112-143
//   FIFO ........................... td4_psp18.py:123-134 (depth 3), td2_psp50.py:98-109 (depth 1)
//   strict state_dict loading ...... td4_psp18.py:232-240
#include "../../include/tdnet.h"
#include "td_device.h"
#include "td_conv.h"
#include "td_conv_h.h"
#include "td_conv_hd.h"
#include "td_conv_ad.h"
#include "td_wino.h"
#include "td_gemm.h"
#include "td_gemm_dma.h"
#include "td_attn.h"
#include "td_attn_h.h"
#include "td_misc.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static int td_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
#define TD_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return td_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define TD_TRY(expr) do { if ((expr) != 0) return -1; } while (0)

// Every C-ABI entry that touches the device runs under the HANDLE's device and restores the caller's current device on exit:
// PyTorch tracks its own current device, and a library that changed it behind torch's back would misplace later allocations;
// a handle on cuda:1 used while the current device is 0 would otherwise launch its kernels and side stream on the wrong GPU.
struct DevGuard {
    int prev = -1;
    bool switched = false, ok = true;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; switched = ok; }
    }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define TD_ON_DEVICE(n, ...)                                                                      \
    DevGuard dev_guard_((n)->cfg.device);                                                         \
    if (!dev_guard_.ok) { td_fail("cannot select HIP device %d", (n)->cfg.device); return __VA_ARGS__; }

extern "C" const char* tdnet_last_error(void) { return g_err; }
// The build stamps the library with a hash of its sources (tdnet_amd/build.py: -DTDNET_SRC_HASH): tests and smoke() compare it with
// the sources that were shipped, so a stale prebuilt .so cannot pass for HEAD's kernels.
#ifndef TDNET_SRC_HASH
#define TDNET_SRC_HASH "unstamped"
#endif
extern "C" const char* tdnet_version(void) { return "tdnet_amd 0.3 (gfx950, fp32 MFMA) tdnet-src-hash:" TDNET_SRC_HASH; }

// ---------------------------------------------------------------------------------------------------------------
// architecture description (same rules as tdnet_amd/arch.py; resnet.py:114-202)
// ---------------------------------------------------------------------------------------------------------------
// bott: conv1x1(cin->planes) conv3x3(planes->planes, stride, dil1) conv1x1(planes->cout); Bottleneck ignores dil2 (resnet.py:62-111)
struct BlockSpec { std::string name; int cin, cout, stride, dil1, dil2; bool ds; bool bott; int planes; };

static std::vector<BlockSpec> backbone_blocks(int backbone) {
    const int nb18[4] = {2, 2, 2, 2}, nb34[4] = {3, 4, 6, 3};
    const int nb101[4] = {3, 4, 23, 3};
    const int* nb = backbone == 18 ? nb18 : backbone == 101 ? nb101 : nb34;   // ResNet-50 has the ResNet-34 block counts
    const bool bott = backbone == 50 || backbone == 101;
    const int exp = bott ? 4 : 1;
    const int planes[4] = {64, 128, 256, 512}, strides[4] = {1, 2, 1, 1}, dils[4] = {1, 1, 2, 4};
    std::vector<BlockSpec> out;
    int inpl = bott ? 128 : 64;                                      // deep_base stem ends in 128 channels (resnet.py:117)
    for (int li = 0; li < 4; ++li) {
        for (int b = 0; b < nb[li]; ++b) {
            const bool first = b == 0, mg = li == 3;
            int d1;
            if (mg) d1 = b == 0 ? 4 : b == 1 ? 8 : 16;            // multi-grid (4,8,16): resnet.py:181,196-198
            else if (first) d1 = (dils[li] == 1 || dils[li] == 2) ? 1 : 2;
            else d1 = dils[li];
            BlockSpec s;
            char nm[32];
            snprintf(nm, sizeof(nm), "layer%d.%d", li + 1, b);
            s.name = nm;
            s.cin = first ? inpl : planes[li] * exp;
            s.cout = planes[li] * exp;
            s.bott = bott;
            s.planes = planes[li];
            s.stride = first ? strides[li] : 1;
            s.dil1 = d1;
            s.dil2 = dils[li];
            s.ds = first && (strides[li] != 1 || inpl != planes[li] * exp);
            out.push_back(s);
        }
        inpl = planes[li] * exp;
    }
    return out;
}
static int feat_size(int n) { for (int i = 0; i < 3; ++i) n = (n - 1) / 2 + 1; return n; }
static int key_size(int n) { return (n - 1) / 4 + 1; }

// ---------------------------------------------------------------------------------------------------------------
// device conv layer
// ---------------------------------------------------------------------------------------------------------------
struct ConvLayer {
    int Cin = 0, Cout = 0, KS = 1, stride = 1, dil = 1, pad = 0, act = 0;
    bool stem = false;
    bool h16 = false;                                                  // fp16-MFMA operands (td_conv_h.h)
    int wino_pad = 0;                                                  // padding rows per Winograd plane (fusion bit 64)
    bool adirect = false;                                              // Cout <= 64: A operand straight from global (td_conv_ad.h, fusion bit 32)
    bool in16 = false, out16 = false;                                  // h16 only: the input (+ residual) / output map is stored as fp16 in HBM
    int rh = 0;                                                        // h16 + in16: != 0 -> the LDS-DMA kernel with 64 rh rows per tile (td_conv_hd.h); M_out: its output pixels
    long M_out = 0;
    bool rowimg_off = false;                                           // tdnet_opts.fusion bit 2048: keep the tap-by-tap LDS-DMA kernel
    int pers = 1;                                                      // tdnet_opts.gemm_persistent of the owning handle
    int stagger = 0;                                                   // tdnet_opts.stagger
    int chunks = 1;                                                    // > 1: run as that many row-parity chunks (tdnet_opts.overlap bit 1); the GEMM tile is picked for T / chunks rows
    bool gdma = false;                                                 // the Winograd GEMMs on the LDS-DMA-fed kernel (td_gemm_dma.h; tdnet_opts.overlap bit 8)
    int vw = 0;                                                        // != 0: the low-register F(4x4) transform kernels with vw channels per lane (td_wino.h k_wino4_*_c)
    int wino = 0;                                                      // Winograd output tile edge m (0 = direct, 2 = F(2x2,3x3), 4 = F(4x4,3x3)): d_wp = (m+2)^2 packed 1x1 weight sets (td_wino.h)
    float* d_zero = nullptr;                                           // zero bias for the batched GEMM pass
    ConvTile tile = CT_128x128;
    int CoutPad = 0, nsteps = 0;
    float* d_wp = nullptr;
    float* d_bias = nullptr;
    double flops_per_pixel() const { return 2.0 * Cout * (stem ? 3.0 * KS * KS : (double)Cin * KS * KS); }
};

// Per-handle kernel configuration (include/tdnet.h tdnet_opts); nothing here is process-wide: two handles in one process may differ.
extern "C" void tdnet_opts_default(tdnet_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->winograd = TDNET_WINOGRAD_DEFAULT;   // F(4x4,3x3) for the stride-1 3x3 convs with Cin, Cout >= 128
    o->precision = 0;                       // fp32 MFMA: the only mode the 1e-3 logits gate applies to
    o->pipeline = 1;                        // two-stage conv prefetch
    o->gemm_persistent = 1;                 // stride-1 1x1 convs and the Winograd GEMMs on the persistent multi-tile GEMM kernel
    o->stagger = 0;
    o->attention = TDNET_ATTENTION_DEFAULT;
    o->fusion = TDNET_FUSION_DEFAULT;
    o->overlap = TDNET_OVERLAP_DEFAULT;
}
static tdnet_opts opts_or_default(const tdnet_opts* o) {
    tdnet_opts d;
    tdnet_opts_default(&d);
    if (!o) return d;
    d = *o;
    d.winograd = d.winograd < 0 ? 0 : d.winograd > 4 ? 4 : d.winograd;
    d.precision = d.precision ? 1 : 0;
    d.pipeline = d.pipeline ? 1 : 0;
    d.gemm_persistent = d.gemm_persistent < 0 ? 0 : d.gemm_persistent;
    d.stagger = d.stagger < 0 ? 0 : d.stagger > 64 ? 64 : d.stagger;
    d.attention = d.attention < 0 ? 0 : d.attention > 2 ? 2 : d.attention;
    d.overlap = d.overlap < 0 ? 0 : d.overlap & 0xff;
    if (((d.overlap >> 4) & 3) == 3) d.overlap &= ~0x30;
    d.cu_reserve = d.cu_reserve < 0 ? 0 : d.cu_reserve > 128 ? 128 : (d.cu_reserve / 8) * 8;
    d.cu_mode &= 7;
    if (!d.cu_reserve) d.cu_mode = 0;
    for (int& r : d.reserved) r = 0;
    return d;
}

static int out_size(int n, int KS, int stride, int dil, int pad) { return (n + 2 * pad - dil * (KS - 1) - 1) / stride + 1; }

// Upload a BN-folded OIHW weight + bias as a ConvLayer for an output of M pixels.
static int make_conv_layer(ConvLayer& L, const std::vector<float>& w, const std::vector<float>& b, int Cout, int Cin, int KS,
                           int stride, int dil, int act, bool stem, long M, const tdnet_opts& o, int forced_tile = -1, int chunks = 1) {
    L.Cin = stem ? 4 : Cin; L.Cout = Cout; L.KS = KS; L.stride = stride; L.dil = dil; L.act = act; L.stem = stem;
    L.pad = stem ? KS / 2 : dil * (KS / 2);
    L.M_out = M;
    L.rowimg_off = (o.fusion & 2048) != 0;
    L.pers = o.gemm_persistent; L.stagger = o.stagger;
    const bool deep = o.pipeline != 0;
    if (!stem && Cin % 32 != 0) return td_fail("conv: Cin=%d is not a multiple of 32", Cin);
    const bool wino_ok = o.winograd && !o.precision && !stem && KS == 3 && stride == 1 && Cin % 32 == 0 && Cout % 4 == 0 &&
                         (o.winograd == 2 || o.winograd == 4 || (Cin >= (o.winograd == 3 ? 128 : 256) && Cout >= 128));
    L.wino = !wino_ok ? 0 : o.winograd >= 3 ? 4 : 2;
    L.wino_pad = (L.wino && (o.fusion & 64) && o.gemm_persistent && gemm_supports(Cin)) ? 24 : 0;   // 24 rows: 12..48 KB between plane phases
    if (L.wino) {
        L.chunks = (L.wino == 4 && chunks > 1 && dil % chunks == 0 && o.gemm_persistent && gemm_supports(Cin)) ? chunks : 1;
        chunks = L.chunks;
        L.vw = (L.wino == 4 && (L.chunks > 1 || (o.overlap & 2))) ? (1 << ((o.overlap >> 4) & 3)) : 0;
        L.gdma = (o.overlap & 8) != 0;
        // nb = (m+2)^2 batched [T x Cin] x [Cin x Cout] GEMMs, T = M / m^2 tiles: nb * T rows in total -> pick the tile for that many workgroups
        const int nb = (L.wino + 2) * (L.wino + 2);
        const bool pers = o.gemm_persistent && gemm_supports(Cin);
        L.tile = forced_tile >= 0 ? (ConvTile)forced_tile
               : pers ? gemm_pick_tile(wino_tiles_estimate(M, dil, L.wino) / chunks, nb, Cout, deep)
                      : conv_pick_tile((int)std::min<long>(nb * M / (L.wino * L.wino), 1 << 30), Cout, deep);
        L.CoutPad = conv_cout_pad(Cout, L.tile);
        L.nsteps = conv_nsteps(Cin, 1, false);
        std::vector<std::vector<float>> U;
        wino_transform_weights(w.data(), Cout, Cin, L.wino, U);
        const size_t per = (size_t)L.nsteps * 8 * L.CoutPad * 4;
        std::vector<float> packed(nb * per);
        for (int bi = 0; bi < nb; ++bi) conv_pack_weights(U[bi].data(), Cout, Cin, 1, false, L.tile, packed.data() + bi * per);
        TD_HIP(hipMalloc((void**)&L.d_wp, packed.size() * sizeof(float)));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        std::vector<float> bb(Cout, 0.f), zz(Cout, 0.f);
        if (!b.empty()) bb = b;
        TD_HIP(hipMalloc((void**)&L.d_bias, Cout * sizeof(float)));
        TD_HIP(hipMemcpy(L.d_bias, bb.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
        TD_HIP(hipMalloc((void**)&L.d_zero, Cout * sizeof(float)));
        TD_HIP(hipMemcpy(L.d_zero, zz.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }
    L.h16 = o.precision && !stem && Cin % 64 == 0;
    const bool stem16 = o.precision && stem && KS == 7 && stride == 2 && Cout <= 64;    // fp16-MFMA stem (td_conv_h.h)
    const bool gemm1x1 = !L.h16 && !stem && KS == 1 && stride == 1 && o.gemm_persistent && gemm_supports(Cin);   // run_conv's persistent-GEMM route
    L.tile = forced_tile >= 0 ? (ConvTile)forced_tile : gemm1x1 ? gemm_pick_tile(M, 1, Cout, deep) : conv_pick_tile((int)M, Cout, deep);
    L.CoutPad = conv_cout_pad(Cout, L.tile);
    if (stem16 && conv_stem_h_supports(L.tile)) {
        L.h16 = true;
        L.nsteps = conv_nsteps_stem_h();
        std::vector<_Float16> packed((size_t)L.nsteps * 8 * L.CoutPad * 8);
        conv_pack_weights_stem_h(w.data(), Cout, L.tile, packed.data());
        TD_HIP(hipMalloc((void**)&L.d_wp, packed.size() * sizeof(_Float16)));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    } else if (L.h16) {
        L.nsteps = conv_nsteps_h(Cin, KS);
        std::vector<_Float16> packed((size_t)L.nsteps * 8 * L.CoutPad * 8);
        conv_pack_weights_h(w.data(), Cout, Cin, KS, L.tile, packed.data());
        TD_HIP(hipMalloc((void**)&L.d_wp, packed.size() * sizeof(_Float16)));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    } else {
        L.nsteps = conv_nsteps(Cin, KS, stem);
        std::vector<float> packed((size_t)L.nsteps * 8 * L.CoutPad * 4);
        conv_pack_weights(w.data(), Cout, Cin, KS, stem, L.tile, packed.data());
        TD_HIP(hipMalloc((void**)&L.d_wp, packed.size() * sizeof(float)));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    L.adirect = (o.fusion & 32) && !L.h16 && !gemm1x1 && conv_adirect_supports(L.tile, 1);
    std::vector<float> bb(Cout, 0.f);
    if (!b.empty()) bb = b;
    TD_HIP(hipMalloc((void**)&L.d_bias, Cout * sizeof(float)));
    TD_HIP(hipMemcpy(L.d_bias, bb.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
static void free_conv_layer(ConvLayer& L) {
    if (L.d_wp) hipFree(L.d_wp);
    if (L.d_bias) hipFree(L.d_bias);
    if (L.d_zero) hipFree(L.d_zero);
    L.d_wp = L.d_bias = L.d_zero = nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------------------
struct BlockLayers { ConvLayer c1, c2, c3, ds; bool has_ds = false, bott = false; };
struct AtnLayer { ConvLayer fc; float* d_bias = nullptr; };          // fc applied to the value matrix (no bias), bias added after P V'
struct PathLayers {
    ConvLayer stem, stem2, stem3;                                      // stem2/3: deep_base only (resnet.py:122-131)
    std::vector<BlockLayers> blocks;
    float* d_ppm_w = nullptr; float* d_ppm_b = nullptr;                // [4][FS][512], [4][FS]
    ConvLayer enc_v, enc_q0, enc_q1, enc_k0, enc_k1;
    std::vector<AtnLayer> atn;                                         // in the order the path applies them
    float* d_ln_g = nullptr; float* d_ln_b = nullptr;                  // [h*w]
    ConvLayer head3;
    float* d_cls_w = nullptr; float* d_cls_b = nullptr;                // [nclass][mid], [nclass]
    int pid = 0;
};
struct CacheSlot { float* q = nullptr; float* k = nullptr; float* v = nullptr; };
struct ProfRec { int family; int dominant; hipEvent_t e0, e1; double flops; };   // dominant: 0 no, 1 direct 3x3 128x128, 2 Winograd batched GEMM

struct tdnet {
    tdnet_cfg cfg;
    tdnet_opts opts;                                                   // per-handle kernel configuration (never process-wide)
    int P = 0, DV = 0, MID = 0, FIFO = 0, C = 512, SC = 64;            // C = backbone output channels, SC = stem output channels
    bool deep = false;
    int H = 0, W = 0, H1 = 0, W1 = 0, H2 = 0, W2 = 0, h = 0, w = 0, hk = 0, wk = 0, Lq = 0, Lk = 0;
    std::vector<BlockSpec> bspec;
    std::map<std::string, std::vector<float>> sd;                      // host state_dict until finalize
    std::map<std::string, size_t> expected;                            // name -> element count
    bool finalized = false;
    std::vector<PathLayers> paths;
    // workspace
    float *img4 = nullptr, *s1 = nullptr, *s1b = nullptr, *bx = nullptr, *bt = nullptr, *br = nullptr, *bu = nullptr;
    float *rowpart = nullptr, *pooled = nullptr, *ppmfeat = nullptr, *z = nullptr;
    float *v_cur = nullptr, *q1 = nullptr, *q_cur = nullptr, *k1 = nullptr;
    float *vp = nullptr, *chain_a = nullptr, *chain_b = nullptr, *feat = nullptr;
    // overlap bit 128: the cache-only chain of the NEXT frame is launched at the end of this one (beside the HBM-bound LayerNorm / head /
    // classifier / upsample instead of beside the next frame's stem and layer1): a second V' buffer, and what the launch assumed
    float* vp2 = nullptr;
    float* vp_read = nullptr;                                          // the V' the final attention of the current frame reads
    bool chain_stale = false;                                          // a pre-launched chain was abandoned and may still be reading cache slots
    bool pre_valid = false;                                            // a chain was pre-launched ...
    int pre_pos = -1;                                                  // ... for this pos_id ...
    unsigned pre_epoch = 0, fifo_epoch = 0;                            // ... with the FIFO as it was at this epoch (reset / external pushes bump it)
    float* pre_vp = nullptr;
    float *ln_part = nullptr, *ln_mean = nullptr, *ln_rstd = nullptr, *ln = nullptr;
    float *headmid = nullptr, *lowres = nullptr, *stage_tmp = nullptr, *logits_tmp = nullptr;
    float *wino_v = nullptr, *wino_m = nullptr;                        // Winograd workspaces [16][T][Cin] / [16][T][Cout]
    size_t wino_v_floats = 0, wino_m_floats = 0;
    size_t stage_tmp_floats = 0;
    std::vector<CacheSlot> slots;
    std::vector<int> fifo;                                             // slot ids, oldest first
    int last_slot = -1;
    int pending_slot = -1;                                             // cache entry of an encoded, not yet propagated frame
    int pending_pos = -1;
    // profiling
    // cache-only work (V' GEMMs + the two cached-frame attention steps) runs on a side stream under the backbone
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;                 // Encoding's q / k projections beside w_vs (fusion bit 1)
    // Row-parity chains (tdnet_opts.overlap bit 1): the trailing run of even-dilation Winograd convs of the backbone starts at conv
    // seg_conv (0: conv1, 1: conv2) of block seg_block (-1: off).  Chain 0 runs on the forward's stream with wino_v / wino_m, chain 1
    // on `chain2` with wino_v2 / wino_m2.  The chains may drift apart by more than a block and the channel count changes inside the
    // run, so no map of the run is written in place or shared between blocks: block b owns seg_t[b] (conv1 output), seg_r[b]
    // (downsample output) and seg_x[b] (block output).
    int seg_block = -1, seg_conv = 0;
    hipStream_t chain2 = nullptr;
    hipEvent_t ev_cfork = nullptr, ev_cjoin = nullptr, ev_cstag = nullptr;
    float *wino_v2 = nullptr, *wino_m2 = nullptr;
    std::vector<float*> seg_t, seg_r, seg_x;
    // tdnet_opts.cu_reserve: the run on a partitioned chip -- part_g (GEMMs of both chains) on all but cu_reserve CUs, part_t (their
    // transforms) on the reserved ones; ev_in[c][i] / ev_g[c][i]: "input transform / GEMMs of chunk c of conv i done"
    std::vector<hipStream_t> probe_streams;                            // TDNET_PROBE_EXTRA_STREAMS (-DTDNET_TIMING_PROBES builds only)
    std::vector<hipStream_t> retired_streams;                          // chain2 candidates that shared the caller's hardware queue (place_chain_stream)
    bool placed = false;
    void* placed_for = nullptr;                                        // the caller stream chain2 was checked against
    int chain_replaced = 0;
    hipStream_t part_g = nullptr, part_t = nullptr;
    std::vector<hipEvent_t> ev_in[2], ev_g[2];
    int part_grid = 0;                                                 // persistent GEMM grid on part_g: 3 workgroups per CU it may use
    float* c4 = nullptr;                                              // backbone output of the last frame (bx, or br in the fp16-activation mode)
    bool act16 = false;                                               // precision = 1: the maps between the backbone's convs are fp16 in HBM
    _Float16* vt16 = nullptr;                                         // fp16 attention: V' transposed [DV][LkPad]
    bool ln_pending = false;                                          // the `ln` map of the last frame was not materialised (fusion bit 4)
    int ln_path = 0;
    bool failed = false;                                              // a launch helper reported an error during the current forward
    bool prof = false;
    std::vector<ProfRec> recs;
    size_t nrec = 0;
    double flops_frame = 0.0;
};

template <typename T>
static int dev_alloc(T** p, size_t count) {
    TD_HIP(hipMalloc((void**)p, count * sizeof(T)));
    return 0;
}

// expected reference state_dict (names + sizes): same inventory as tdnet_amd/arch.py:state_dict_shapes
static void add_bn(std::map<std::string, size_t>& e, const std::string& pre, int c) {
    e[pre + ".weight"] = c; e[pre + ".bias"] = c; e[pre + ".running_mean"] = c; e[pre + ".running_var"] = c;
    e[pre + ".num_batches_tracked"] = 1;
}
static std::vector<std::string> atn_module_names(int model) {
    if (model == 4) return {"atn1_2", "atn1_3", "atn1_4", "atn2_1", "atn2_3", "atn2_4", "atn3_1", "atn3_2", "atn3_4", "atn4_1", "atn4_2", "atn4_3"};
    return {"atn1", "atn2"};
}
// attention modules in application order for path p (0-based): td4_psp18.py:145-147,166-168,185-187,204-206
static std::vector<std::string> atn_order(int model, int p) {
    if (model == 2) return {p == 0 ? "atn1" : "atn2"};
    static const char* t[4][3] = {{"atn1_2", "atn1_3", "atn1_4"}, {"atn2_3", "atn2_4", "atn2_1"},
                                  {"atn3_4", "atn3_1", "atn3_2"}, {"atn4_1", "atn4_2", "atn4_3"}};
    return {t[p][0], t[p][1], t[p][2]};
}
static void build_expected(tdnet* n) {
    auto& e = n->expected;
    char b[160];
    for (int p = 1; p <= n->P; ++p) {
        snprintf(b, sizeof(b), "pretrained%d", p);
        std::string pre = n->cfg.model == 1 ? std::string("pretrained") : std::string(b);   // pspnet.py:51-64: self.pretrained
        if (n->deep) {
            e[pre + ".conv1.0.weight"] = 64 * 3 * 9; add_bn(e, pre + ".conv1.1", 64);
            e[pre + ".conv1.3.weight"] = 64 * 64 * 9; add_bn(e, pre + ".conv1.4", 64);
            e[pre + ".conv1.6.weight"] = 128 * 64 * 9; add_bn(e, pre + ".bn1", 128);
        } else {
            e[pre + ".conv1.weight"] = 64 * 3 * 49;
            add_bn(e, pre + ".bn1", 64);
        }
        for (auto& s : n->bspec) {
            std::string bp = pre + "." + s.name;
            if (s.bott) {
                e[bp + ".conv1.weight"] = (size_t)s.planes * s.cin; add_bn(e, bp + ".bn1", s.planes);
                e[bp + ".conv2.weight"] = (size_t)s.planes * s.planes * 9; add_bn(e, bp + ".bn2", s.planes);
                e[bp + ".conv3.weight"] = (size_t)s.cout * s.planes; add_bn(e, bp + ".bn3", s.cout);
            } else {
                e[bp + ".conv1.weight"] = (size_t)s.cout * s.cin * 9; add_bn(e, bp + ".bn1", s.cout);
                e[bp + ".conv2.weight"] = (size_t)s.cout * s.cout * 9; add_bn(e, bp + ".bn2", s.cout);
            }
            if (s.ds) { e[bp + ".downsample.0.weight"] = (size_t)s.cout * s.cin; add_bn(e, bp + ".downsample.1", s.cout); }
        }
        e[pre + ".fc.weight"] = (size_t)1000 * n->C; e[pre + ".fc.bias"] = 1000;
        if (n->cfg.model == 1) {                                       // PSPHead: pspnet.py:102-115
            for (int j = 1; j <= 4; ++j) {
                snprintf(b, sizeof(b), "head.conv5.0.conv%d", j);
                e[std::string(b) + ".0.weight"] = (size_t)(n->C / 4) * n->C;
                add_bn(e, std::string(b) + ".1", n->C / 4);
            }
            e["head.conv5.1.weight"] = (size_t)(n->C / 4) * 2 * n->C * 9; add_bn(e, "head.conv5.2", n->C / 4);
            e["head.conv5.5.weight"] = (size_t)n->cfg.nclass * (n->C / 4); e["head.conv5.5.bias"] = n->cfg.nclass;
            continue;
        }
        for (int j = 1; j <= 4; ++j) {
            snprintf(b, sizeof(b), "psp%d.conv%d", p, j);
            e[std::string(b) + ".0.weight"] = (size_t)(n->C / 4) * n->C;
            add_bn(e, std::string(b) + ".1", n->C / 4);
        }
        for (const char* br : {"w_qs", "w_ks"}) {
            snprintf(b, sizeof(b), "enc%d.%s", p, br);
            std::string ep = b;
            e[ep + ".0.conv.weight"] = (size_t)64 * n->C; e[ep + ".0.conv.bias"] = 64; add_bn(e, ep + ".0.bn", 64);
            e[ep + ".1.conv.weight"] = 64 * 64; e[ep + ".1.conv.bias"] = 64;
        }
        snprintf(b, sizeof(b), "enc%d.w_vs.0.conv", p);
        e[std::string(b) + ".weight"] = (size_t)n->DV * n->C; e[std::string(b) + ".bias"] = n->DV;
        snprintf(b, sizeof(b), "layer_norm%d.ln", p);
        e[std::string(b) + ".weight"] = (size_t)n->h * n->w; e[std::string(b) + ".bias"] = (size_t)n->h * n->w;
        snprintf(b, sizeof(b), "head%d.conv5", p);
        std::string hp = b;
        e[hp + ".0.weight"] = (size_t)n->MID * n->DV * 9; add_bn(e, hp + ".1", n->MID);
        e[hp + ".4.weight"] = (size_t)n->cfg.nclass * n->MID; e[hp + ".4.bias"] = n->cfg.nclass;
    }
    if (n->cfg.model != 1)
    for (auto& a : atn_module_names(n->cfg.model)) {
        e[a + ".fc.0.conv.weight"] = (size_t)n->DV * n->DV; e[a + ".fc.0.conv.bias"] = n->DV;
    }
}

extern "C" int tdnet_create(const tdnet_cfg* cfg, tdnet_t** out) { return tdnet_create_opts(cfg, nullptr, out); }

extern "C" int tdnet_create_opts(const tdnet_cfg* cfg, const tdnet_opts* opts, tdnet_t** out) {
    if (!cfg || !out) return td_fail("tdnet_create: null argument");
    if (cfg->model != 4 && cfg->model != 2 && cfg->model != 1)
        return td_fail("tdnet_create: model must be 4 (td4), 2 (td2) or 1 (single-frame PSPNet), got %d", cfg->model);
    if (cfg->backbone != 18 && cfg->backbone != 34 && cfg->backbone != 50 && cfg->backbone != 101)
        return td_fail("tdnet_create: backbone must be 18, 34, 50 or 101, got %d", cfg->backbone);
    if ((cfg->backbone == 101) != (cfg->model == 1) && !(cfg->model == 1 && cfg->backbone == 50))
        return td_fail("tdnet_create: resnet101 is the PSPNet baseline's backbone (pspnet.py:36); psp accepts 50 or 101");
    if (cfg->nclass < 1 || cfg->nclass > 32) return td_fail("tdnet_create: nclass must be in 1..32");
    if (cfg->height < 9 || cfg->width < 9) return td_fail("tdnet_create: input too small");
    {
        int ndev = 0;
        TD_HIP(hipGetDeviceCount(&ndev));
        if (cfg->device < 0 || cfg->device >= ndev) return td_fail("tdnet_create: device %d out of range (%d visible)", cfg->device, ndev);
    }
    tdnet* n = new tdnet();
    n->cfg = *cfg;
    n->opts = opts_or_default(opts);
    n->P = cfg->model;
    const int exp = cfg->backbone >= 50 ? 4 : 1;                       // Bottleneck expansion (td2_psp50.py:63-66)
    n->deep = cfg->backbone >= 50;
    n->C = 512 * exp;
    n->SC = n->deep ? 128 : 64;
    n->DV = cfg->model == 4 ? 512 * exp : 128 * exp;                   // td4_psp18.py:85 (512*expansion) / td2_psp50.py:79 (512*exp//4)
    n->MID = cfg->model == 4 ? n->DV / 4 : n->DV / 2;                  // FCNHead chn_down 4 / 2
    if (cfg->model == 1) { n->DV = 2 * n->C; n->MID = n->C / 4; }      // PSPHead: conv3x3 on the 2C-channel concat -> C/4 (pspnet.py:105-109)
    n->FIFO = cfg->model == 4 ? 3 : cfg->model == 2 ? 1 : 0;
    n->H = cfg->height; n->W = cfg->width;
    n->H1 = (n->H - 1) / 2 + 1; n->W1 = (n->W - 1) / 2 + 1;
    n->H2 = (n->H1 - 1) / 2 + 1; n->W2 = (n->W1 - 1) / 2 + 1;
    n->h = feat_size(n->H); n->w = feat_size(n->W);
    n->hk = key_size(n->h); n->wk = key_size(n->w);
    n->Lq = n->h * n->w; n->Lk = n->hk * n->wk;
    n->bspec = backbone_blocks(cfg->backbone);
    build_expected(n);
    *out = n;
    return 0;
}

static void free_path(PathLayers& p) {
    free_conv_layer(p.stem); free_conv_layer(p.stem2); free_conv_layer(p.stem3);
    for (auto& b : p.blocks) { free_conv_layer(b.c1); free_conv_layer(b.c2); free_conv_layer(b.c3); free_conv_layer(b.ds); }
    for (ConvLayer* c : {&p.enc_v, &p.enc_q0, &p.enc_q1, &p.enc_k0, &p.enc_k1, &p.head3}) free_conv_layer(*c);
    for (auto& a : p.atn) { free_conv_layer(a.fc); if (a.d_bias) hipFree(a.d_bias); }
    for (float* q : {p.d_ppm_w, p.d_ppm_b, p.d_ln_g, p.d_ln_b, p.d_cls_w, p.d_cls_b}) if (q) hipFree(q);
}
extern "C" void tdnet_destroy(tdnet_t* n) {
    if (!n) return;
    TD_ON_DEVICE(n);
    for (auto& p : n->paths) free_path(p);
    for (float* q : {n->img4, n->s1, n->s1b, n->bx, n->bt, n->br, n->bu, n->rowpart, n->pooled, n->ppmfeat, n->z, n->v_cur, n->q1, n->q_cur,
                     n->k1, n->vp, n->vp2, n->chain_a, n->chain_b, n->feat, n->ln_part, n->ln_mean, n->ln_rstd, n->ln, n->headmid,
                     n->lowres, n->stage_tmp, n->logits_tmp, n->wino_v, n->wino_m})
        if (q) hipFree(q);
    for (auto& s : n->slots) { if (s.q) hipFree(s.q); if (s.k) hipFree(s.k); if (s.v) hipFree(s.v); }
    for (auto& r : n->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    if (n->vt16) hipFree(n->vt16);
    for (float* q : {n->wino_v2, n->wino_m2}) if (q) hipFree(q);
    for (auto* v : {&n->seg_t, &n->seg_r, &n->seg_x}) for (float* q : *v) if (q) hipFree(q);
    if (n->chain2) hipStreamDestroy(n->chain2);
    for (hipStream_t x : n->probe_streams) hipStreamDestroy(x);
    for (hipStream_t x : n->retired_streams) hipStreamDestroy(x);
    if (n->part_g) hipStreamDestroy(n->part_g);
    if (n->part_t) hipStreamDestroy(n->part_t);
    for (auto* v : {&n->ev_in[0], &n->ev_in[1], &n->ev_g[0], &n->ev_g[1]}) for (hipEvent_t e : *v) if (e) hipEventDestroy(e);
    if (n->ev_cfork) hipEventDestroy(n->ev_cfork);
    if (n->ev_cjoin) hipEventDestroy(n->ev_cjoin);
    if (n->ev_cstag) hipEventDestroy(n->ev_cstag);
    if (n->side) hipStreamDestroy(n->side);
    if (n->ev_fork) hipEventDestroy(n->ev_fork);
    if (n->ev_join) hipEventDestroy(n->ev_join);
    if (n->ev_fork2) hipEventDestroy(n->ev_fork2);
    if (n->ev_join2) hipEventDestroy(n->ev_join2);
    delete n;
}

extern "C" int tdnet_set_weight(tdnet_t* n, const char* name, const float* host, size_t count) {
    if (!n || !name || !host) return td_fail("tdnet_set_weight: null argument");
    if (n->finalized) return td_fail("tdnet_set_weight: weights already finalized");
    auto it = n->expected.find(name);
    if (it == n->expected.end()) return td_fail("Unexpected key in state_dict: \"%s\"", name);
    if (it->second != count) return td_fail("size mismatch for %s: expected %zu elements, got %zu", name, it->second, count);
    const std::string s = name;
    if (s.find(".fc.weight") != std::string::npos && s.compare(0, 10, "pretrained") == 0) return 0;   // unused classifier
    if (s.find(".fc.bias") != std::string::npos && s.compare(0, 10, "pretrained") == 0) return 0;
    if (s.size() > 19 && s.compare(s.size() - 19, 19, "num_batches_tracked") == 0) { n->sd[s] = {0.f}; return 0; }
    n->sd[s].assign(host, host + count);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BN folding (fp64): y = (conv(x)+b - mu) * g / sqrt(var + eps) + beta   (td4_psp18.py:23-24)
// ---------------------------------------------------------------------------------------------------------------
struct Folded { std::vector<float> w, b; };
static const std::vector<float>& T(tdnet* n, const std::string& k) { return n->sd.at(k); }
static Folded fold(tdnet* n, const std::string& wkey, const std::string& bkey, const std::string& bn, int Cout) {
    Folded f;
    const std::vector<float>& w = T(n, wkey);
    const size_t per = w.size() / Cout;
    f.w.resize(w.size());
    f.b.assign(Cout, 0.f);
    for (int o = 0; o < Cout; ++o) {
        double scale = 1.0, shift = 0.0, cb = bkey.empty() ? 0.0 : (double)T(n, bkey)[o];
        if (!bn.empty()) {
            const double g = T(n, bn + ".weight")[o], be = T(n, bn + ".bias")[o], mu = T(n, bn + ".running_mean")[o],
                         var = T(n, bn + ".running_var")[o];
            scale = g / std::sqrt(var + 1e-5);
            shift = be - mu * scale;
        }
        for (size_t i = 0; i < per; ++i) f.w[o * per + i] = (float)((double)w[o * per + i] * scale);
        f.b[o] = (float)(cb * scale + shift);
    }
    return f;
}
static int upload(float** d, const std::vector<float>& v) {
    TD_HIP(hipMalloc((void**)d, v.size() * sizeof(float)));
    TD_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// Row-parity chains: which convs can run as an even-row and an odd-row half (tdnet_opts.overlap bit 1).  A stride-1 3x3 conv with an
// EVEN dilation reads, for an output row y, only the input rows y + k * dil: rows of y's parity.  So from the first such conv to the end
// of the backbone (ResNet-18/34: layer3.0.conv2 .. layer4.1.conv2, dilations 2,2,2,4,4,8,4 -- resnet.py:181-198) the even and the odd
// rows are two INDEPENDENT chains of convs (residual adds and the 1x1 downsample are pixel-wise), each with half the Winograd tiles.
static bool conv_chainable(int cin, int cout, int stride, int dil, const tdnet_opts& o) {
    // rider schedule (overlap bit 64): a chunk's 36 GEMM batches go out as 24 + 12, which are whole tiles per resident workgroup only
    // when a batch of a chunk is a multiple of 64 tiles of 64 x 128 -- the 512-channel layers; the narrower ones stay outside the run
    if ((o.overlap & 64) && (o.overlap & 8) && cout < 512) return false;
    return stride == 1 && dil % 2 == 0 && cin >= 128 && cout >= 128 && gemm_supports(cin) && cout % 4 == 0;
}
static void plan_chains(tdnet* n) {
    n->seg_block = -1; n->seg_conv = 0;
    const tdnet_opts& o = n->opts;
    if (!(o.overlap & 1) || o.winograd < 3 || o.precision || !o.gemm_persistent || n->deep) return;
    int sb = -1, sc = 0;
    for (int b = (int)n->bspec.size() - 1; b >= 0; --b) {
        const BlockSpec& S = n->bspec[b];
        if (S.bott || !conv_chainable(S.cout, S.cout, 1, S.dil2, o)) break;
        sb = b; sc = 1;
        if (!conv_chainable(S.cin, S.cout, S.stride, S.dil1, o)) break;
        sb = b; sc = 0;
        if (S.ds && !(S.stride == 1 && gemm_supports(S.cin))) break;   // an earlier start would put this block's downsample inside the chains
    }
    n->seg_block = sb; n->seg_conv = sc;
}
static bool in_chain(const tdnet* n, int block, int conv) {
    return n->seg_block >= 0 && (block > n->seg_block || (block == n->seg_block && conv >= n->seg_conv));
}

static int alloc_workspace(tdnet* n) {
    const size_t hw = (size_t)n->Lq, lk = (size_t)n->Lk;
    const bool psp = n->cfg.model == 1;
    const size_t C = n->C, FS = psp ? C / 4 : C / 8, ZC = psp ? 2 * C : C;
    size_t bmax = (size_t)n->H2 * n->W2 * n->SC, cmax = (size_t)n->H2 * n->W2 * 64;   // bmax: block in/out, cmax: inner (planes) maps
    {
        int ch = n->H2, cw = n->W2;
        for (auto& s : n->bspec) {
            const int oh = out_size(ch, 3, s.stride, s.dil1, s.dil1), ow = out_size(cw, 3, s.stride, s.dil1, s.dil1);
            bmax = std::max(bmax, (size_t)oh * ow * s.cout);
            cmax = std::max(cmax, (size_t)ch * cw * (s.bott ? s.planes : s.cout));     // Bottleneck conv1 output is at the INPUT resolution
            ch = oh; cw = ow;
        }
    }
    if (dev_alloc(&n->img4, (size_t)n->H * n->W * 4)) return -1;
    if (dev_alloc(&n->s1, (size_t)n->H1 * n->W1 * 64)) return -1;
    if (n->deep && dev_alloc(&n->s1b, (size_t)n->H1 * n->W1 * 64)) return -1;
    if (n->deep) bmax = std::max(bmax, (size_t)n->H1 * n->W1 * 128);   // br also holds the deep stem output
    if (dev_alloc(&n->bx, bmax) || dev_alloc(&n->br, bmax) || dev_alloc(&n->bt, std::max(cmax, n->bspec[0].bott ? (size_t)0 : bmax))) return -1;
    if (n->deep && dev_alloc(&n->bu, cmax)) return -1;
    if (dev_alloc(&n->rowpart, (size_t)n->h * 36 * C) || dev_alloc(&n->pooled, 50 * C) || dev_alloc(&n->ppmfeat, 50 * FS)) return -1;
    if (dev_alloc(&n->z, hw * ZC)) return -1;
    n->stage_tmp_floats = hw * ZC;
    if (dev_alloc(&n->headmid, hw * n->MID) || dev_alloc(&n->lowres, hw * n->cfg.nclass) || dev_alloc(&n->stage_tmp, n->stage_tmp_floats)) return -1;
    {   // Winograd workspaces: the largest [(m+2)^2][T][C] over the layers that use it (all paths share them; one stream)
        size_t vmax = 0, mmax = 0;
        auto upd = [&](const ConvLayer& L, int H, int W) {
            if (!L.wino) return;
            const size_t T = (size_t)wino_tiles(H, W, L.dil, L.wino), nb = (size_t)(L.wino + 2) * (L.wino + 2);
            vmax = std::max(vmax, nb * (T + L.wino_pad) * L.Cin); mmax = std::max(mmax, nb * (T + L.wino_pad) * L.Cout);
        };
        const PathLayers& L0 = n->paths[0];
        if (n->deep) { upd(L0.stem2, n->H1, n->W1); upd(L0.stem3, n->H1, n->W1); }
        int ch = n->H2, cw = n->W2;
        for (size_t i = 0; i < L0.blocks.size(); ++i) {
            const BlockSpec& bs = n->bspec[i];
            const int oh = out_size(ch, 3, bs.stride, bs.dil1, bs.dil1), ow = out_size(cw, 3, bs.stride, bs.dil1, bs.dil1);
            if (bs.bott) upd(L0.blocks[i].c2, ch, cw);
            else { upd(L0.blocks[i].c1, ch, cw); upd(L0.blocks[i].c2, oh, ow); }
            ch = oh; cw = ow;
        }
        upd(L0.head3, n->h, n->w);
        n->wino_v_floats = vmax; n->wino_m_floats = mmax;
        if (vmax && (dev_alloc(&n->wino_v, vmax) || dev_alloc(&n->wino_m, mmax))) return -1;
        if (n->seg_block >= 0) {                                       // chain 1's own workspaces (a chunk is never larger than the conv) + the run's maps
            if (dev_alloc(&n->wino_v2, vmax) || dev_alloc(&n->wino_m2, mmax)) return -1;
            const size_t nb = L0.blocks.size();
            n->seg_t.assign(nb, nullptr); n->seg_r.assign(nb, nullptr); n->seg_x.assign(nb, nullptr);
            for (size_t b = (size_t)n->seg_block; b < nb; ++b) {
                const size_t sz = (size_t)n->h * n->w * n->bspec[b].cout;   // the run is at the backbone's output resolution (stride-1 convs)
                if (dev_alloc(&n->seg_t[b], sz) || dev_alloc(&n->seg_x[b], sz)) return -1;
                if (n->bspec[b].ds && dev_alloc(&n->seg_r[b], sz)) return -1;
            }
        }
    }
    if (psp) return 0;
    if (dev_alloc(&n->v_cur, hw * n->DV) || dev_alloc(&n->q1, hw * 64) || dev_alloc(&n->q_cur, hw * 64)) return -1;
    if (dev_alloc(&n->k1, lk * 64) || dev_alloc(&n->vp, (size_t)attn_vp_rows((int)lk) * n->DV) || dev_alloc(&n->chain_a, lk * n->DV) || dev_alloc(&n->chain_b, lk * n->DV)) return -1;
    TD_HIP(hipMemset(n->vp, 0, (size_t)attn_vp_rows((int)lk) * n->DV * sizeof(float)));   // padding rows of V' stay zero (td_attn.h load_v)
    if (n->opts.overlap & 128) {
        if (dev_alloc(&n->vp2, (size_t)attn_vp_rows((int)lk) * n->DV)) return -1;
        TD_HIP(hipMemset(n->vp2, 0, (size_t)attn_vp_rows((int)lk) * n->DV * sizeof(float)));
    }
    n->vp_read = n->vp;
    if (dev_alloc(&n->feat, hw * n->DV) || dev_alloc(&n->ln, hw * n->DV)) return -1;
    const size_t ln_strips = std::max<size_t>(512, (size_t)attn_strips(n->Lq, n->DV));   // k_ln_stats: <= 512 strips; attention epilogue: one per query tile
    if (dev_alloc(&n->ln_part, 2 * ln_strips * n->DV) || dev_alloc(&n->ln_mean, n->DV) || dev_alloc(&n->ln_rstd, n->DV)) return -1;
    if (n->opts.precision && hipMalloc((void**)&n->vt16, (size_t)n->DV * attn_lkpad((int)lk) * sizeof(_Float16)) != hipSuccess)
        return td_fail("hipMalloc failed for the fp16 attention workspace");
    n->slots.resize(n->FIFO + 2);                                      // FIFO + the pending entry + one being received
    for (auto& s : n->slots)
        if (dev_alloc(&s.q, lk * 64) || dev_alloc(&s.k, lk * 64) || dev_alloc(&s.v, lk * n->DV)) return -1;
    return 0;
}

static double frame_flops(const tdnet* n);

extern "C" int tdnet_finalize_weights(tdnet_t* n) {
    if (!n) return td_fail("tdnet_finalize_weights: null handle");
    if (n->finalized) return td_fail("tdnet_finalize_weights: already finalized");
    TD_ON_DEVICE(n, -1);
    for (auto& kv : n->expected) {
        const std::string& k = kv.first;
        if (k.compare(0, 10, "pretrained") == 0 && (k.find(".fc.weight") != std::string::npos || k.find(".fc.bias") != std::string::npos)) continue;
        if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) continue;
        if (!n->sd.count(k)) return td_fail("Missing key in state_dict: \"%s\"", k.c_str());
    }
    const int C = n->C, DV = n->DV, FS = C / (2 * 4), NC = n->cfg.nclass;
    plan_chains(n);
    char b[160];
    // The row-parity plan (plan_chains / conv_chainable) and the per-layer decision (make_conv_layer: Winograd F(4x4) + chunkable) are two
    // predicates over the same facts.  Should they ever disagree, the chains are a schedule, not a requirement: the layers are rebuilt
    // unchained (attempt 1) instead of failing the load.
    for (int attempt = 0; attempt < 2; ++attempt) {
    bool plan_mismatch = false;
    for (auto& pl : n->paths) free_path(pl);
    n->paths.clear();
    n->paths.resize(n->P);
    for (int p = 0; p < n->P && !plan_mismatch; ++p) {
        PathLayers& L = n->paths[p];
        L.pid = p & 1;                                                 // td4_psp18.py:80-83 / td2_psp50.py:76-77
        snprintf(b, sizeof(b), "pretrained%d", p + 1);
        const std::string pre = n->cfg.model == 1 ? std::string("pretrained") : std::string(b);
        if (n->deep) {                                                 // conv3x3 s2 3->64, conv3x3 64->64, conv3x3 64->128 (+bn1)
            Folded f0 = fold(n, pre + ".conv1.0.weight", "", pre + ".conv1.1", 64);
            if (make_conv_layer(L.stem, f0.w, f0.b, 64, 3, 3, 2, 1, 1, true, (long)n->H1 * n->W1, n->opts)) return -1;
            Folded f1 = fold(n, pre + ".conv1.3.weight", "", pre + ".conv1.4", 64);
            if (make_conv_layer(L.stem2, f1.w, f1.b, 64, 64, 3, 1, 1, 1, false, (long)n->H1 * n->W1, n->opts)) return -1;
            Folded f2 = fold(n, pre + ".conv1.6.weight", "", pre + ".bn1", 128);
            if (make_conv_layer(L.stem3, f2.w, f2.b, 128, 64, 3, 1, 1, 1, false, (long)n->H1 * n->W1, n->opts)) return -1;
        } else {
            Folded f = fold(n, pre + ".conv1.weight", "", pre + ".bn1", 64);
            if (make_conv_layer(L.stem, f.w, f.b, 64, 3, 7, 2, 1, 1, true, (long)n->H1 * n->W1, n->opts)) return -1;
        }
        int ch = n->H2, cw = n->W2;
        for (size_t bsi = 0; bsi < n->bspec.size(); ++bsi) {
            const BlockSpec& s = n->bspec[bsi];
            const int k1 = in_chain(n, (int)bsi, 0) ? 2 : 1, k2 = in_chain(n, (int)bsi, 1) ? 2 : 1;   // row-parity chunks of conv1 / conv2
            BlockLayers B;
            const std::string bp = pre + "." + s.name;
            const int oh = out_size(ch, 3, s.stride, s.dil1, s.dil1), ow = out_size(cw, 3, s.stride, s.dil1, s.dil1);
            const long M = (long)oh * ow;
            B.bott = s.bott;
            if (s.bott) {
                Folded f1 = fold(n, bp + ".conv1.weight", "", bp + ".bn1", s.planes);
                if (make_conv_layer(B.c1, f1.w, f1.b, s.planes, s.cin, 1, 1, 1, 1, false, (long)ch * cw, n->opts)) return -1;
                Folded f2 = fold(n, bp + ".conv2.weight", "", bp + ".bn2", s.planes);
                if (make_conv_layer(B.c2, f2.w, f2.b, s.planes, s.planes, 3, s.stride, s.dil1, 1, false, M, n->opts)) return -1;
                Folded f3 = fold(n, bp + ".conv3.weight", "", bp + ".bn3", s.cout);
                if (make_conv_layer(B.c3, f3.w, f3.b, s.cout, s.planes, 1, 1, 1, 1, false, M, n->opts)) return -1;   // ReLU after the residual add
            } else {
                Folded f1 = fold(n, bp + ".conv1.weight", "", bp + ".bn1", s.cout);
                if (make_conv_layer(B.c1, f1.w, f1.b, s.cout, s.cin, 3, s.stride, s.dil1, 1, false, M, n->opts, -1, k1)) return -1;
                Folded f2 = fold(n, bp + ".conv2.weight", "", bp + ".bn2", s.cout);
                if (make_conv_layer(B.c2, f2.w, f2.b, s.cout, s.cout, 3, 1, s.dil2, 1, false, M, n->opts, -1, k2)) return -1;
                if ((k1 > 1 && B.c1.chunks != k1) || (k2 > 1 && B.c2.chunks != k2)) plan_mismatch = true;
            }
            B.has_ds = s.ds;
            if (s.ds) {
                Folded fd = fold(n, bp + ".downsample.0.weight", "", bp + ".downsample.1", s.cout);
                if (make_conv_layer(B.ds, fd.w, fd.b, s.cout, s.cin, 1, s.stride, 1, 0, false, M, n->opts)) return -1;
            }
            L.blocks.push_back(B);
            ch = oh; cw = ow;
            if (plan_mismatch) break;
        }
        if (plan_mismatch) break;
        if (ch != n->h || cw != n->w) return td_fail("internal: feature size mismatch %dx%d vs %dx%d", ch, cw, n->h, n->w);
        if (n->cfg.model == 1) {                                       // PSPHead (pspnet.py:102-115): full pyramid, conv3x3, classifier
            const int F4 = C / 4;
            std::vector<float> pw((size_t)4 * F4 * C), pb((size_t)4 * F4);
            for (int j = 0; j < 4; ++j) {
                snprintf(b, sizeof(b), "head.conv5.0.conv%d", j + 1);
                Folded f = fold(n, std::string(b) + ".0.weight", "", std::string(b) + ".1", F4);
                for (int o = 0; o < F4; ++o) {
                    for (int c = 0; c < C; ++c) pw[((size_t)j * C + c) * F4 + o] = f.w[(size_t)o * C + c];
                    pb[j * F4 + o] = f.b[o];
                }
            }
            if (upload(&L.d_ppm_w, pw) || upload(&L.d_ppm_b, pb)) return -1;
            Folded fh = fold(n, "head.conv5.1.weight", "", "head.conv5.2", n->MID);
            if (make_conv_layer(L.head3, fh.w, fh.b, n->MID, 2 * C, 3, 1, 1, 1, false, n->Lq, n->opts)) return -1;
            if (upload(&L.d_cls_w, T(n, "head.conv5.5.weight")) || upload(&L.d_cls_b, T(n, "head.conv5.5.bias"))) return -1;
            continue;
        }
        // pyramid convs: keep only the FS output channels this path's slice uses (td4_psp18.py:279-282)
        std::vector<float> pw((size_t)4 * FS * C), pb((size_t)4 * FS);
        for (int j = 0; j < 4; ++j) {
            snprintf(b, sizeof(b), "psp%d.conv%d", p + 1, j + 1);
            Folded f = fold(n, std::string(b) + ".0.weight", "", std::string(b) + ".1", C / 4);
            for (int o = 0; o < FS; ++o) {
                for (int c = 0; c < C; ++c) pw[((size_t)j * C + c) * FS + o] = f.w[(size_t)(L.pid * FS + o) * C + c];   // [lvl][c][f]
                pb[j * FS + o] = f.b[L.pid * FS + o];
            }
        }
        if (upload(&L.d_ppm_w, pw) || upload(&L.d_ppm_b, pb)) return -1;
        snprintf(b, sizeof(b), "enc%d", p + 1);
        const std::string ep = b;
        {
            Folded fv = fold(n, ep + ".w_vs.0.conv.weight", ep + ".w_vs.0.conv.bias", "", DV);
            if (make_conv_layer(L.enc_v, fv.w, fv.b, DV, C, 1, 1, 1, 0, false, n->Lq, n->opts)) return -1;
            Folded q0 = fold(n, ep + ".w_qs.0.conv.weight", ep + ".w_qs.0.conv.bias", ep + ".w_qs.0.bn", 64);
            if (make_conv_layer(L.enc_q0, q0.w, q0.b, 64, C, 1, 1, 1, 2, false, n->Lq, n->opts)) return -1;
            Folded q1 = fold(n, ep + ".w_qs.1.conv.weight", ep + ".w_qs.1.conv.bias", "", 64);
            if (make_conv_layer(L.enc_q1, q1.w, q1.b, 64, 64, 1, 1, 1, 0, false, n->Lq, n->opts)) return -1;
            Folded k0 = fold(n, ep + ".w_ks.0.conv.weight", ep + ".w_ks.0.conv.bias", ep + ".w_ks.0.bn", 64);
            if (make_conv_layer(L.enc_k0, k0.w, k0.b, 64, C, 1, 4, 1, 2, false, n->Lk, n->opts)) return -1;   // stride 4 = the key sub-sampling
            Folded k1 = fold(n, ep + ".w_ks.1.conv.weight", ep + ".w_ks.1.conv.bias", "", 64);
            if (make_conv_layer(L.enc_k1, k1.w, k1.b, 64, 64, 1, 1, 1, 0, false, n->Lk, n->opts)) return -1;
        }
        for (auto& an : atn_order(n->cfg.model, p)) {
            AtnLayer A;
            std::vector<float> nob;
            if (make_conv_layer(A.fc, T(n, an + ".fc.0.conv.weight"), nob, DV, DV, 1, 1, 1, 0, false, n->Lk, n->opts)) return -1;
            if (upload(&A.d_bias, T(n, an + ".fc.0.conv.bias"))) return -1;
            L.atn.push_back(A);
        }
        snprintf(b, sizeof(b), "layer_norm%d.ln", p + 1);
        if (upload(&L.d_ln_g, T(n, std::string(b) + ".weight")) || upload(&L.d_ln_b, T(n, std::string(b) + ".bias"))) return -1;
        snprintf(b, sizeof(b), "head%d.conv5", p + 1);
        const std::string hp = b;
        Folded fh = fold(n, hp + ".0.weight", "", hp + ".1", n->MID);
        if (make_conv_layer(L.head3, fh.w, fh.b, n->MID, DV, 3, 1, 1, 1, false, n->Lq, n->opts)) return -1;
        if (upload(&L.d_cls_w, T(n, hp + ".4.weight")) || upload(&L.d_cls_b, T(n, hp + ".4.bias"))) return -1;
        (void)NC;
    }
    if (!plan_mismatch) break;
    if (attempt == 1) return td_fail("internal: conv layers ask for row-parity chunks without a chain plan");
    n->seg_block = -1; n->seg_conv = 0;                                 // rebuild every layer with chunks = 1
    }
    // precision = 1: every map between two convs of the backbone is stored as fp16 (half the conv input / output bytes; td_conv_h.h).
    // The rim: the 7x7 stem runs on the fp16 MFMA from the fp32 image and writes an fp16 map (the 3x3 deep stem's first conv stays an
    // fp32 kernel with an fp32 map), and c4 -- the LAST conv of the backbone -- writes fp32 for the pyramid, Encoding and head
    // kernels, which keep fp32 storage.
    n->act16 = n->opts.precision != 0;
    if (n->act16)
        for (auto& L : n->paths) {
            bool all16 = true;
            for (auto& B : L.blocks) all16 = all16 && B.c1.h16 && B.c2.h16 && (!B.bott || B.c3.h16) && (!B.has_ds || B.ds.h16);
            if (n->deep) all16 = all16 && L.stem2.h16 && L.stem3.h16;
            all16 = all16 && !L.blocks.empty() && !L.blocks.back().has_ds;
            if (!all16) { n->act16 = false; break; }
        }
    if (n->act16)
        for (auto& L : n->paths) {
            if (n->deep) { L.stem2.out16 = true; L.stem3.in16 = L.stem3.out16 = true; }
            else if (L.stem.h16) L.stem.out16 = true;                  // fp16-MFMA 7x7 stem: its map is fp16 too (max-pool reads fp16)
            for (size_t bi = 0; bi < L.blocks.size(); ++bi) {
                BlockLayers& B = L.blocks[bi];
                const bool last = bi + 1 == L.blocks.size();
                B.c1.in16 = B.c1.out16 = true;
                B.c2.in16 = true;
                if (B.bott) { B.c2.out16 = true; B.c3.in16 = true; B.c3.out16 = !last; }
                else B.c2.out16 = !last;
                if (B.has_ds) B.ds.in16 = B.ds.out16 = true;
            }
            // fp16 maps in, Cout >= 128: the LDS-DMA kernel (td_conv_hd.h), unless fusion bit 128 keeps the register-staged one
            auto dma = [&](ConvLayer& c) {
                if (!c.h16 || !c.in16 || c.stem || (n->opts.fusion & 128)) return;
                if (c.Cout >= 128 && conv_dma_supports(c.Cin, c.Cout, c.KS, c.tile)) {
                    c.rh = conv_dma_pick_rh(c.M_out, c.Cout, c.CoutPad % 256 == 0 && !(n->opts.fusion & 1024));
                    // Small maps (720x960: 10800 output pixels): a 3x3 "same" conv with <= 256 output channels on NARROW tiles (rows x 64
                    // channels, k_conv_dma_h3n) -- half the weight bytes per K step and CU, the term that dominates there: 128 channels
                    // 13.8 -> 10.3 us, 256 channels 20.6 -> 19.8 us isolated (profiles/r04u_*).  No gain at 32768 pixels.
                    const bool same3 = c.KS == 3 && c.stride == 1 && c.pad == c.dil;
                    if ((n->opts.fusion & 32768) && same3 && c.M_out <= 16384 && c.Cout <= 256)
                        c.rh = c.Cout <= 128 ? CD_128_N : CD_192_N;       // (256 channels on 128 x 64 tiles as well: 2.1 % instead of 2.6 % in the frame)
                    else if (n->opts.fusion & 8192)                     // the 128- and 192-row tiles with four dedicated loader waves (k_conv_dma_h3p):
                        c.rh = c.rh == CD_128 ? CD_128_P : c.rh == CD_192 ? CD_192_P : c.rh;
                        // isolated 22.5 -> 21.0 / 57.8 -> 56.7 us; 256 rows: no gain (profiles/r04d_*).  (192 rows with TWELVE matrix waves of 32 x 64 -- three
                        // per SIMD instead of two SIMDs with twice the MFMAs -- is 3-5 % faster alone and 0.6 % SLOWER in the frame: profiles/r04z_*.)
                }
                else if ((n->opts.fusion & 4096) && conv_dma_w64_supports(c.Cin, c.Cout, c.CoutPad, c.KS, c.stride, c.dil, c.pad))
                    c.rh = CD_W64;                                      // layer1: weights resident in LDS (k_conv_dma_w64; measured no faster, opt-in)
            };
            if (n->deep) dma(L.stem3);
            for (auto& B : L.blocks) { dma(B.c1); dma(B.c2); if (B.bott) dma(B.c3); if (B.has_ds) dma(B.ds); }
            // The head's 3x3 conv (d_v -> d_v / 4 channels; >= 128 for td4): LayerNorm writes its map as fp16 -- the rounding the conv
            // applied to the fp32 map while staging it -- and the conv runs on the LDS-DMA kernel (1024x2048: 77 -> 46 us).
            if (n->cfg.model != 1 && L.head3.h16 && !L.head3.wino) { L.head3.in16 = true; dma(L.head3); if (!L.head3.rh) L.head3.in16 = false; }
        }
    {   // A frame runs on three hardware queues at once; with HIP's default of 4 queues per priority class a process that creates a few
        // streams of its own (torch's pools, RCCL) owns enough queues that they are no longer all resident, and frames run at 0.66x
        // (tdnet_amd/__init__.py, profiles/r04l_*).  The Python package sets GPU_MAX_HW_QUEUES=2 before the runtime starts; a C caller is told once.
        static bool told = false;
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (!told && !getenv("TDNET_QUIET") && (!q || atoi(q) > 3)) {
            told = true;
            fprintf(stderr, "tdnet: GPU_MAX_HW_QUEUES is %s; export GPU_MAX_HW_QUEUES=2 before the HIP runtime starts -- with more hardware queues "
                            "alive in the process this handle's streams are time-sliced (275 -> 185 frames/s measured behind an RCCL communicator)\n", q ? q : "not set");
        }
    }
    n->sd.clear();
    if (alloc_workspace(n)) return -1;
    TD_HIP(hipDeviceSynchronize());
#ifdef TDNET_TIMING_PROBES
    if (const char* e = getenv("TDNET_PROBE_EXTRA_STREAMS")) {          // probe builds only: k extra streams before the handle's own shift its queue placement (DESIGN_experiments 8.4)
        for (int i = 0, k = atoi(e); i < k && i < 16; ++i) {
            hipStream_t x = nullptr;
            if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) == hipSuccess) n->probe_streams.push_back(x);
        }
    }
#endif
    {   // The side stream carries the cache-only attention chain (0.6 ms of work beside 2.5 ms of backbone): lowest priority, so its
        // workgroups fill what the critical path leaves instead of taking CUs from it.
        int least = 0, greatest = 0;
        TD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        TD_HIP(hipStreamCreateWithPriority(&n->side, hipStreamNonBlocking, least));
    }
    if (n->seg_block >= 0 && n->opts.cu_reserve > 0 && !((n->opts.overlap & 64) && (n->opts.overlap & 8))) {
        // Two hardware queues with DISJOINT compute-unit sets.  A queue's CU mask is a bit vector over the device's CUs; the driver deals
        // bit i to XCD i mod 8 (then round-robin over that XCD's shader engines), so the low R bits are R / 8 CUs of every XCD.
        int ncu = 0;
        TD_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, n->cfg.device));
        const int R = n->opts.cu_reserve, words = (ncu + 31) / 32;
        if (ncu < 64 || R >= ncu) return td_fail("cu_reserve %d on a device with %d compute units", R, ncu);
        std::vector<uint32_t> mt((size_t)words, 0u), mg((size_t)words, 0u);
        for (int i = 0; i < ncu; ++i) {
            const bool reserved = (n->opts.cu_mode & 2) ? (i % 32) < R / 8 && i / 32 < 8 : i < R;
            (reserved ? mt : mg)[i / 32] |= 1u << (i % 32);
        }
        if (n->opts.cu_mode & 4) TD_HIP(hipStreamCreateWithFlags(&n->part_g, hipStreamNonBlocking));   // diagnostic: the pipeline without masks
        else TD_HIP(hipExtStreamCreateWithCUMask(&n->part_g, (uint32_t)words, mg.data()));
        if (n->opts.cu_mode & 5) TD_HIP(hipStreamCreateWithFlags(&n->part_t, hipStreamNonBlocking));
        else TD_HIP(hipExtStreamCreateWithCUMask(&n->part_t, (uint32_t)words, mt.data()));
        n->part_grid = 3 * (ncu - R);
        const size_t nev = 2 * n->bspec.size() + 2;
        for (int c = 0; c < 2; ++c) {
            n->ev_in[c].assign(nev, nullptr); n->ev_g[c].assign(nev, nullptr);
            for (size_t i = 0; i < nev; ++i) {
                TD_HIP(hipEventCreateWithFlags(&n->ev_in[c][i], hipEventDisableTiming));
                TD_HIP(hipEventCreateWithFlags(&n->ev_g[c][i], hipEventDisableTiming));
            }
        }
    }
    if (n->seg_block >= 0) {
        // (Which hardware queue this stream gets is HIP's choice; place_chain_stream() checks it against the caller's at the first frame.)
        TD_HIP(hipStreamCreateWithFlags(&n->chain2, hipStreamNonBlocking));
        TD_HIP(hipEventCreateWithFlags(&n->ev_cfork, hipEventDisableTiming));
        TD_HIP(hipEventCreateWithFlags(&n->ev_cjoin, hipEventDisableTiming));
        TD_HIP(hipEventCreateWithFlags(&n->ev_cstag, hipEventDisableTiming));
    }
    TD_HIP(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
    TD_HIP(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
    TD_HIP(hipEventCreateWithFlags(&n->ev_fork2, hipEventDisableTiming));
    TD_HIP(hipEventCreateWithFlags(&n->ev_join2, hipEventDisableTiming));
    n->finalized = true;
    n->flops_frame = frame_flops(n);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------------------------
// TIMING PROBE, compiled in only with -DTDNET_TIMING_PROBES (TDNET_EXTRA_CXXFLAGS of tdnet_amd/build.py; never in the shipped library): under
// TDNET_PROBE_SKIP=<mask> a handle leaves pieces of the frame OUT -- the results are garbage -- to measure what the frame costs without them:
//   1 = no Winograd transforms, 2 = no cache-only attention chain, 4 = no final attention, 8 = no Winograd GEMMs.  Upper bounds for what any
//   optimisation of that piece can return (DESIGN_experiments 8.6, profiles/r04z_frame_budget_*).
#ifdef TDNET_TIMING_PROBES
static int probe_skip() {
    static const int m = [] {
        const char* e = getenv("TDNET_PROBE_SKIP");
        const int v = e ? atoi(e) : 0;
        if (v) fprintf(stderr, "tdnet: TDNET_PROBE_SKIP=%d -- pieces of the frame are NOT computed, every result of this process is garbage (timing probe build)\n", v);
        return v;
    }();
    return m;
}
#else
static constexpr int probe_skip() { return 0; }
#endif
static void prof_begin(tdnet* n, int family, int dominant, double flops, hipStream_t s) {
    if (!n || !n->prof) return;
    if (n->nrec == n->recs.size()) {
        ProfRec r;
        hipEventCreate(&r.e0); hipEventCreate(&r.e1);
        n->recs.push_back(r);
    }
    ProfRec& r = n->recs[n->nrec];
    r.family = family; r.dominant = dominant; r.flops = flops;
    hipEventRecord(r.e0, s);
}
static void prof_end(tdnet* n, hipStream_t s) {
    if (!n || !n->prof) return;
    hipEventRecord(n->recs[n->nrec].e1, s);
    n->nrec++;
}

// plane LayerNorm applied to the conv's INPUT inside a Winograd input transform (td_wino.h WinoArgs.ln_*)
struct LnFuse { const float *mean, *rstd, *g, *b; };

// One row-parity / column-parity chunk of a Winograd conv (td_wino.h WinoArgs.Tc..cx): the tiles whose phase row is ny * i + cy and
// whose phase column is nx * j + cx.  {1, 0, 1, 0} = the whole conv.
struct WinoChunk { int ny = 1, cy = 0, nx = 1, cx = 0; };

template <int VW>
static void launch_wino4_c(bool out_side, const WinoArgs& wa, hipStream_t s) {
    if (out_side) TD_LAUNCH((k_wino4_out_c<VW>), dim3(wino_chunk_grid(wa.Tc, wa.Cout, VW)), dim3(256), 0, s, wa);
    else TD_LAUNCH((k_wino4_in_c<VW>), dim3(wino_chunk_grid(wa.Tc, wa.C, VW)), dim3(256), 0, s, wa);
}

// Winograd conv (or one chunk of it): input transform -> (m+2)^2 batched GEMMs -> output transform, all on stream s.
// V / Mb: workspaces for THIS call ([nb][Tc + pad][C]); nullptr = the handle's (n->wino_v / wino_m) or, without a handle, temporary ones.
// after_in / waiter: an event recorded right after the input transform and a stream made to wait for it (the staggered start of the
// second row-parity chain); nullptr = none.
static int run_wino(tdnet* n, const ConvLayer& L, const float* in, int H, int W, const float* resid, float* out, hipStream_t s,
                    const LnFuse* lnf, const WinoChunk& ck, float* V, float* Mb, hipEvent_t after_in = nullptr, hipStream_t waiter = nullptr) {
    const int TY = wino_tiles_1d(H, L.dil, L.wino), TX = wino_tiles_1d(W, L.dil, L.wino);
    const long T = (long)L.dil * L.dil * TY * TX;
    const bool chunked = ck.ny != 1 || ck.nx != 1;
    if (chunked && (!L.vw || L.dil % ck.ny || L.dil % ck.nx)) return td_fail("internal: this conv cannot run in chunks");
    const long Tc = (long)(L.dil / ck.ny) * (L.dil / ck.nx) * TY * TX;
    const long TP = Tc + L.wino_pad;                                   // padded plane (td_wino.h WinoArgs.TP)
    const int nb = (L.wino + 2) * (L.wino + 2);
    bool own = false;
    if (!V) {
        own = n == nullptr || n->wino_v_floats < (size_t)nb * TP * L.Cin || n->wino_m_floats < (size_t)nb * TP * L.Cout;
        if (own) {
            if (n) { n->failed = true; return td_fail("internal: Winograd workspace too small"); }
            if (dev_alloc(&V, (size_t)nb * TP * L.Cin) || dev_alloc(&Mb, (size_t)nb * TP * L.Cout)) return -1;
        } else { V = n->wino_v; Mb = n->wino_m; }
    }
    WinoArgs wa;
    wa.in = in; wa.V = V; wa.Mb = Mb; wa.bias = L.d_bias; wa.resid = resid; wa.out = out;
    wa.H = H; wa.W = W; wa.C = L.Cin; wa.Cout = L.Cout; wa.dil = L.dil; wa.TY = TY; wa.TX = TX; wa.T = (int)T; wa.act = L.act; wa.TP = (int)TP;
    wa.ln_mean = lnf ? lnf->mean : nullptr; wa.ln_rstd = lnf ? lnf->rstd : nullptr; wa.ln_g = lnf ? lnf->g : nullptr; wa.ln_b = lnf ? lnf->b : nullptr;
    wa.Tc = (int)Tc; wa.ny = ck.ny; wa.cy = ck.cy; wa.nx = ck.nx; wa.cx = ck.cx;
    auto transform = [&](bool out_side) {
        if (n && (probe_skip() & 1)) return;
        prof_begin(n, 2, false, 0, s);
        const int C = out_side ? L.Cout : L.Cin;
        if (L.vw == 1) launch_wino4_c<1>(out_side, wa, s);
        else if (L.vw == 2 && C % 2 == 0) launch_wino4_c<2>(out_side, wa, s);
        else if (L.vw == 4 && C % 4 == 0) launch_wino4_c<4>(out_side, wa, s);
        else if (L.vw) launch_wino4_c<1>(out_side, wa, s);
        else if (L.wino == 4 && out_side) TD_LAUNCH(k_wino4_out, dim3(td_grid_for(T * (L.Cout / 4), 256, 256 * 16)), dim3(256), 0, s, wa);
        else if (L.wino == 4) TD_LAUNCH(k_wino4_in, dim3(td_grid_for(T * (L.Cin / 4), 256, 256 * 16)), dim3(256), 0, s, wa);
        else if (out_side) TD_LAUNCH(k_wino_out, dim3(td_grid_for(T * (L.Cout / 4), 256, 256 * 16)), dim3(256), 0, s, wa);
        else TD_LAUNCH(k_wino_in, dim3(td_grid_for(T * (L.Cin / 4), 256, 256 * 16)), dim3(256), 0, s, wa);
        prof_end(n, s);
    };
    transform(false);
    if (after_in && waiter) {
        TD_HIP(hipEventRecord(after_in, s));
        TD_HIP(hipStreamWaitEvent(waiter, after_in, 0));
    }
    prof_begin(n, 0, 2, 2.0 * nb * Tc * (double)L.Cin * L.Cout, s);
    if (n && (probe_skip() & 8)) { /* timing probe: no GEMMs */ }
    else if (L.pers && gemm_supports(L.Cin)) {
        GemmArgs ga;
        ga.a = V; ga.wp = L.d_wp; ga.bias = L.d_zero; ga.resid = nullptr; ga.out = Mb;
        ga.M = (int)Tc; ga.N = L.Cout; ga.NPad = L.CoutPad; ga.K = L.Cin; ga.nbatch = nb; ga.act = 0; ga.tiles_m = ga.tiles_n = 0; ga.MP = (int)TP; ga.stagger = L.stagger;
        if (L.gdma && gemm_dma_supports(L.Cin, L.Cout, L.tile)) gemm_dma_launch(ga, nullptr, L.pers > 1 ? L.pers : 0, s);
        else gemm_launch(ga, L.tile, L.pers > 1 ? L.pers : 0, s);
    } else {
        ConvArgs g;
        g.in = V; g.wp = L.d_wp; g.bias = L.d_zero; g.resid = nullptr; g.out = Mb;
        g.H = 1; g.W = (int)Tc; g.Cin = L.Cin; g.Wo = (int)Tc; g.Cout = L.Cout; g.CoutPad = L.CoutPad;
        g.stride = 1; g.dil = 1; g.pad = 0; g.M = (int)Tc; g.nsteps = L.nsteps; g.act = 0; g.tiles_n = 0; g.stagger = 0; g.nbatch = nb;
        conv_launch(g, L.tile, 1, false, s);
    }
    prof_end(n, s);
    transform(true);
    if (own) { TD_HIP(hipStreamSynchronize(s)); hipFree(V); hipFree(Mb); }
    return 0;
}

// out[Ho*Wo][Cout] = act(conv(in[H][W][Cin]) + bias (+ resid))
static int run_conv(tdnet* n, const ConvLayer& L, const float* in, int H, int W, const float* resid, float* out, hipStream_t s,
                    int* Ho_out = nullptr, int* Wo_out = nullptr, const LnFuse* lnf = nullptr) {
    if (lnf && !L.wino) return td_fail("internal: LayerNorm fusion needs a Winograd input transform");
    const int Ho = out_size(H, L.KS, L.stride, L.dil, L.pad), Wo = out_size(W, L.KS, L.stride, L.dil, L.pad);
    if (L.wino) {
        if (Ho_out) *Ho_out = H;
        if (Wo_out) *Wo_out = W;
        if (L.chunks > 1 && !n) {                                      // operator tests: the chunks one after the other on one stream
            for (int c = 0; c < L.chunks; ++c) {
                WinoChunk ck; ck.ny = L.chunks; ck.cy = c;
                TD_TRY(run_wino(n, L, in, H, W, resid, out, s, lnf, ck, nullptr, nullptr));
            }
            return 0;
        }
        return run_wino(n, L, in, H, W, resid, out, s, lnf, WinoChunk(), nullptr, nullptr);
    }
    ConvArgs a;
    a.in = in; a.wp = L.d_wp; a.bias = L.d_bias; a.resid = resid; a.out = out;
    a.H = H; a.W = W; a.Cin = L.Cin; a.Wo = Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.stride = L.stride; a.dil = L.dil; a.pad = L.pad; a.M = Ho * Wo; a.nsteps = L.nsteps; a.act = L.act; a.tiles_n = 0; a.stagger = L.stagger; a.nbatch = 1;
    prof_begin(n, 0, (L.tile == CT_128x128 || L.tile == CT_128x128_DEEP || (L.rh && L.rh != CD_W64)) && L.KS == 3 && !L.stem, L.flops_per_pixel() * a.M, s);
    if (L.h16 && L.stem) conv_launch_stem_h(a, L.out16, s);
    else if (L.h16 && L.rh == CD_W64) {                                 // 64 -> 64 channels: persistent workgroups with the weights resident in LDS
        if (!conv_launch_dma_w64(a, L.out16, s)) conv_launch_h(a, L.tile, L.KS, L.in16, L.out16, s);
    } else if (L.h16 && L.rh) {                                         // 3x3 stride 1: one LDS image per kernel ROW (k_conv_dma_h3) where the halo fits
        // A cascade of kernel forms for the SAME tile, each falling back to the next when the conv does not qualify (not a 3x3 "same" conv,
        // halo wider than the form's image buffer): narrow tiles -> loader waves -> row images -> tap by tap.
        int rh = L.rh;
        bool done = false;
        const bool is192 = rh == CD_192_P || rh == CD_192_N, is256 = rh == CD_256_P || rh == CD_256_N;
        if (rh == CD_128_N || rh == CD_192_N || rh == CD_256_N) {        // narrow tiles (rows x 64 channels) with loader waves (k_conv_dma_h3n)
            done = !L.rowimg_off && conv_launch_dma3n(a, rh, L.KS, L.out16, s);
            if (!done) rh = is192 ? CD_192_P : is256 ? CD_256_P : CD_128_P;
        }
        if (!done && (rh == CD_128_P || rh == CD_192_P || rh == CD_256_P)) {                              // dedicated loader waves (k_conv_dma_h3p)
            done = !L.rowimg_off && conv_launch_dma3p(a, rh, L.KS, L.out16, s);
            if (!done) rh = is192 ? CD_192 : is256 ? CD_256 : CD_128_8W;
        }
        if (!done && (L.rowimg_off || !conv_launch_dma3(a, rh, L.KS, L.out16, s))) conv_launch_dma(a, rh, L.KS, L.out16, s);
    }
    else if (L.h16) conv_launch_h(a, L.tile, L.KS, L.in16, L.out16, s);
    else if (L.pers && L.KS == 1 && L.stride == 1 && !L.stem && gemm_supports(L.Cin)) {
        GemmArgs ga;
        ga.a = in; ga.wp = L.d_wp; ga.bias = L.d_bias; ga.resid = resid; ga.out = out;
        ga.M = a.M; ga.N = L.Cout; ga.NPad = L.CoutPad; ga.K = L.Cin; ga.nbatch = 1; ga.act = L.act; ga.tiles_m = ga.tiles_n = 0; ga.MP = a.M; ga.stagger = L.stagger;
        gemm_launch(ga, L.tile, L.pers > 1 ? L.pers : 0, s);
    } else if (L.adirect) conv_launch_adirect(a, L.KS, L.stem, s);
    else conv_launch(a, L.tile, L.KS, L.stem, s);
    prof_end(n, s);
    if (Ho_out) *Ho_out = Ho;
    if (Wo_out) *Wo_out = Wo;
    return 0;
}

// ln_part != nullptr: the kernel also writes the plane-LayerNorm strip statistics of `out` (one strip per 32-row query tile)
static int run_attention(tdnet* n, const float* q, const float* k, const float* vp, const float* bias, const float* resid,
                         int Lq, int Lk, int DV, float* out, hipStream_t s, int online = 0, float* ln_part = nullptr,
                         _Float16* vt16 = nullptr, bool slices = false) {
    if (n && n->vt16) vt16 = n->vt16;
    AttnArgs a;
    a.q = q; a.k = k; a.vp = vp; a.bias = bias; a.resid = resid; a.out = out; a.Lq = Lq; a.Lk = Lk;
    a.scale_log2e = 1.4426950408889634f / 8.0f;                        // temperature = sqrt(d_k) = 8 (transformer.py:65)
    a.ln_part = ln_part; a.ln_nstr = 0;
    if (n && (probe_skip() & 4) && Lq > Lk) return 0;
    prof_begin(n, 1, false, 2.0 * Lq * (double)Lk * (64 + DV), s);
    const int rc = vt16 ? attn_launch_h(a, DV, vt16, s) : attn_launch(a, DV, online, s, slices);   // vt16: the fp16-MFMA kernel (tdnet_opts.precision = 1)
    prof_end(n, s);
    if (rc) return td_fail("attention: unsupported d_v=%d (128 or a multiple of 512)", DV);
    return 0;
}

// Plane LayerNorm (td4_psp18.py:306-312) in up to three launches: strip statistics (skipped when the attention epilogue already
// wrote them: stats_nstr > 0 strips of 32 rows), their exact combination, and the normalisation (skipped when y == nullptr: the
// head's Winograd input transform applies it on the fly, run_conv's LnFuse).
static void run_layernorm(tdnet* n, const float* x, int HW, int C, const float* g, const float* b, float* part, float* mean,
                          float* rstd, float* y, hipStream_t s, int stats_nstr = 0, bool y16 = false) {
    const int CV = C / 4, threads = CV > 256 ? CV : 256, rows = threads / CV;                   // C = 2048 (td4 on ResNet-50): 512 threads, one row each
    int nstr = stats_nstr, per = 32;
    prof_begin(n, 2, false, 0, s);
    if (!stats_nstr) {
        nstr = (HW + rows - 1) / rows;
        if (nstr > 512) nstr = 512;
        per = (HW + nstr - 1) / nstr;                                                           // k_ln_stats' strip length
        TD_LAUNCH(k_ln_stats, dim3(nstr), dim3(threads), (rows + 1) * C * 4, s, x, part, HW, C);   // part: [2][nstr][C]
    }
    TD_LAUNCH(k_ln_finalize, dim3((C + 3) / 4), dim3(256), (256 + 32 + 4) * 4, s, (const float*)part, nstr, per, HW, C, 1e-5f, mean, rstd);
    if (y && y16) TD_LAUNCH(k_ln_apply_h, dim3(td_grid_for((long)HW * CV)), dim3(256), 0, s, x, (const float*)mean, (const float*)rstd, g, b, (_Float16*)y, HW, C);
    else if (y) TD_LAUNCH(k_ln_apply, dim3(td_grid_for((long)HW * CV)), dim3(256), 0, s, x, (const float*)mean, (const float*)rstd, g, b, y, HW, C);
    prof_end(n, s);
}

// XS = channels of c4 kept (c/path_num, offset pid*XS), FS = channels kept of each pyramid conv (c/(4 path_num))
static void run_ppm(tdnet* n, const float* c4, int h, int w, int C, int XS, int FS, const float* wgt, const float* bias, int pid,
                    float* rowpart, float* pooled, float* ppmfeat, float* z, hipStream_t s) {
    prof_begin(n, 2, false, 0, s);
    const PpmAtoms at = ppm_atoms(w);                                 // the row is read once: atoms between the bin edges of all four levels
    TD_LAUNCH(k_ppm_rowsum, dim3(h * at.n), dim3(C / 4), 0, s, c4, rowpart, w, C, at);
    float* rowbins = rowpart + (size_t)h * 24 * C;                    // [h][12][C] behind the (at most 23) atoms per row
    TD_LAUNCH(k_ppm_rowbins, dim3(h * 12), dim3(C / 4), 0, s, (const float*)rowpart, rowbins, C, at);
    TD_LAUNCH(k_ppm_bins, dim3(50), dim3(C / 4), 0, s, (const float*)rowbins, pooled, h, w, C);
    TD_LAUNCH(k_ppm_conv, dim3(50 * (FS / 64)), dim3(256), 256 * 4, s, (const float*)pooled, wgt, bias, ppmfeat, C, FS);
    TD_LAUNCH(k_ppm_assemble, dim3(td_grid_for((long)h * w * (C / 4))), dim3(256), 0, s, c4, (const float*)ppmfeat, z, h, w, C,
              pid * XS, XS, FS);
    prof_end(n, s);
}

static void run_stem_pre(tdnet* n, const float* img, int H, int W, float* img4, hipStream_t s, int fusion) {
    prof_begin(n, 2, false, 0, s);
    if ((fusion & (16 | 256)) && (H * W) % 4 == 0 && ((size_t)img & 15) == 0)
        TD_LAUNCH(k_nchw3_to_nhwc4_x4, dim3(td_grid_for((long)H * W / 4)), dim3(256), 0, s, img, img4, H * W);
    else
    TD_LAUNCH(k_nchw3_to_nhwc4, dim3(td_grid_for((long)H * W)), dim3(256), 0, s, img, img4, H * W);
    prof_end(n, s);
}
// pool16: 0 = fp32 in / fp32 out; the fp16-activation mode's first map: 1 = fp32 in (the stem's output) / fp16 out, 2 = fp16 in (deep stem) / fp16 out
static void run_maxpool(tdnet* n, const float* in, int H, int W, int C, float* out, hipStream_t s, int fusion, int pool16 = 0) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    prof_begin(n, 2, false, 0, s);
    if (pool16) {
        if (pool16 == 2) TD_LAUNCH((k_maxpool3s2_h<true>), dim3(td_grid_for((long)Ho * Wo * (C / 4))), dim3(256), 0, s, (const void*)in, (_Float16*)out, H, W, C, Ho, Wo);
        else TD_LAUNCH((k_maxpool3s2_h<false>), dim3(td_grid_for((long)Ho * Wo * (C / 4))), dim3(256), 0, s, (const void*)in, (_Float16*)out, H, W, C, Ho, Wo);
    } else if (fusion & 16)
        TD_LAUNCH(k_maxpool3s2_x2, dim3(td_grid_for((long)Ho * ((Wo + 1) / 2) * (C / 4), 256, 256 * 16)), dim3(256), 0, s, in, out, H, W, C, Ho, Wo);
    else
    TD_LAUNCH(k_maxpool3s2, dim3(td_grid_for((long)Ho * Wo * (C / 4))), dim3(256), 0, s, in, out, H, W, C, Ho, Wo);
    prof_end(n, s);
}
static int run_classifier(tdnet* n, const float* x, int HW, int C, int NC, const float* wgt, const float* bias, float* out, hipStream_t s) {
    if (C % 16) return td_fail("classifier: C=%d is not a multiple of 16", C);
    prof_begin(n, 2, false, 0, s);
    const int grid = (HW + 63) / 64, lds = (NC * C + 4 * NC * 64) * 4;
    if (NC <= 19) TD_LAUNCH((k_classifier<19>), dim3(grid), dim3(256), lds, s, x, wgt, bias, out, HW, C, NC);
    else TD_LAUNCH((k_classifier<32>), dim3(grid), dim3(256), lds, s, x, wgt, bias, out, HW, C, NC);
    prof_end(n, s);
    return 0;
}

static void launch_upsample(const float* in, int C, int h, int w, int H, int W, float* out, hipStream_t s) {
    if (W % 4 == 0 && H <= 65535 && C <= 65535) TD_LAUNCH(k_upsample_x4, dim3((W / 4 + 255) / 256, H, C), dim3(256), 0, s, in, out, C, h, w, H, W);
    else TD_LAUNCH(k_upsample, dim3(td_grid_for((long)C * H * W, 256, 256 * 16)), dim3(256), 0, s, in, out, C, h, w, H, W);
}

// low-resolution logits of one frame (planar [nclass][h*w]) + FIFO update
// A frame is three pieces (td4_psp18.py:137-154):
//   chain    everything that depends only on CACHED frames (:145-146 and the fc of :147) -- side stream, joined before the
//            final attention;
//   encode   backbone, pyramid slice, Encoding(pre=False) and Encoding(pre=True): ends with q_cur / v_cur and this frame's own
//            cache entry in a PENDING slot (not yet in the FIFO);
//   finish   final attention against the newest cached frame, plane LayerNorm, head, classifier; then the pending entry is
//            committed to the FIFO (:153-154, :123-134).
// tdnet_forward runs chain || encode, then finish.  tdnet_encode / tdnet_propagate expose the two halves so that a
// path-parallel deployment can exchange cache entries between them.
static int free_slot(tdnet* n) {
    for (int i = 0; i < (int)n->slots.size(); ++i) {
        bool used = i == n->pending_slot;
        for (int f : n->fifo) used |= f == i;
        if (!used) return i;
    }
    return -1;
}
static void fifo_commit(tdnet* n, int slot) {
    n->fifo.push_back(slot);
    if ((int)n->fifo.size() > n->FIFO) n->fifo.erase(n->fifo.begin());
    n->last_slot = slot;
}

// vp: the V' buffer the chain ends in (the final attention of the frame it belongs to reads it); e0, e1, e2: the cache slots that frame
// sees as its FIFO, oldest first (td2: e0 only).
static int launch_chain(tdnet* n, PathLayers& L, hipStream_t s, float* vp, int e0, int e1, int e2) {
    const int DV = n->DV;
    hipStream_t c = n->side;
    TD_HIP(hipEventRecord(n->ev_fork, s));
    TD_HIP(hipStreamWaitEvent(c, n->ev_fork, 0));
    if (probe_skip() & 2) { TD_HIP(hipEventRecord(n->ev_join, c)); return 0; }
    if (n->P == 4) {
        const CacheSlot &c0 = n->slots[e0], &c1 = n->slots[e1], &c2 = n->slots[e2];
        TD_TRY(run_conv(n, L.atn[0].fc, c0.v, 1, n->Lk, nullptr, vp, c));
        // the cached-frame steps have Lq = Lk (64 query tiles at 1024x2048): two channel slices per launch unless fusion bit 512 says no
        const bool sl = !(n->opts.fusion & 512) && DV == 512 && n->Lk <= 8192;
        if (run_attention(n, c1.q, c0.k, vp, L.atn[0].d_bias, c1.v, n->Lk, n->Lk, DV, n->chain_a, c, n->opts.attention, nullptr, nullptr, sl)) return -1;   // v2 + V[1]
        TD_TRY(run_conv(n, L.atn[1].fc, n->chain_a, 1, n->Lk, nullptr, vp, c));
        if (run_attention(n, c2.q, c1.k, vp, L.atn[1].d_bias, c2.v, n->Lk, n->Lk, DV, n->chain_b, c, n->opts.attention, nullptr, nullptr, sl)) return -1;   // v3 + V[2]
        TD_TRY(run_conv(n, L.atn[2].fc, n->chain_b, 1, n->Lk, nullptr, vp, c));                                              // (v3 + V[2]) W^T
    } else {
        TD_TRY(run_conv(n, L.atn[0].fc, n->slots[e0].v, 1, n->Lk, nullptr, vp, c));
    }
    TD_HIP(hipEventRecord(n->ev_join, c));
    return 0;
}
// the chain of the frame that is about to be propagated, against the FIFO as it stands
static int launch_chain_now(tdnet* n, PathLayers& L, hipStream_t s) {
    n->vp_read = n->vp;
    return launch_chain(n, L, s, n->vp, n->fifo[0], n->P == 4 ? n->fifo[1] : -1, n->P == 4 ? n->fifo[2] : -1);
}

static int run_ds_rows(tdnet* n, const ConvLayer& L, const float* in, int H, int W, float* out, int ny, int cy, hipStream_t s);

// ---- the pieces of a Winograd conv chunk for the rider schedule (td_gemm_dma.h): its argument block, a stand-alone transform launch,
// and the GEMMs of the batches [b0, b0 + nbs) with the transforms of OTHER chunks riding in the fifth wave of its workgroups -------------
static WinoArgs wino_chunk_args(const ConvLayer& L, const float* in, int H, int W, const float* resid, float* out, const WinoChunk& ck,
                                float* V, float* Mb) {
    const int TY = wino_tiles_1d(H, L.dil, L.wino), TX = wino_tiles_1d(W, L.dil, L.wino);
    const long Tc = (long)(L.dil / ck.ny) * (L.dil / ck.nx) * TY * TX;
    WinoArgs wa;
    wa.in = in; wa.V = V; wa.Mb = Mb; wa.bias = L.d_bias; wa.resid = resid; wa.out = out;
    wa.H = H; wa.W = W; wa.C = L.Cin; wa.Cout = L.Cout; wa.dil = L.dil; wa.TY = TY; wa.TX = TX; wa.T = (int)((long)L.dil * L.dil * TY * TX);
    wa.act = L.act; wa.TP = (int)(Tc + L.wino_pad);
    wa.ln_mean = wa.ln_rstd = wa.ln_g = wa.ln_b = nullptr;
    wa.Tc = (int)Tc; wa.ny = ck.ny; wa.cy = ck.cy; wa.nx = ck.nx; wa.cx = ck.cx;
    return wa;
}
static void wino_transform_alone(tdnet* n, const ConvLayer& L, const WinoArgs& wa, bool out_side, hipStream_t s) {
    if (n && (probe_skip() & 1)) return;
    prof_begin(n, 2, false, 0, s);
    const int C = out_side ? L.Cout : L.Cin;
    if (L.vw == 4 && C % 4 == 0) launch_wino4_c<4>(out_side, wa, s);
    else if (L.vw == 2 && C % 2 == 0) launch_wino4_c<2>(out_side, wa, s);
    else launch_wino4_c<1>(out_side, wa, s);
    prof_end(n, s);
}
static void wino_gemm_with_rider(tdnet* n, const ConvLayer& L, const WinoArgs& wa, int b0, int nbs, const WinoArgs* ride_in,
                                 const WinoArgs* ride_out, hipStream_t s) {
    GemmArgs ga;
    const int nsteps = L.Cin / 32;
    ga.a = wa.V + (size_t)b0 * wa.TP * L.Cin; ga.wp = L.d_wp + (size_t)b0 * nsteps * 8 * L.CoutPad * 4; ga.bias = L.d_zero; ga.resid = nullptr;
    ga.out = const_cast<float*>(wa.Mb) + (size_t)b0 * wa.TP * L.Cout;
    ga.M = wa.Tc; ga.N = L.Cout; ga.NPad = L.CoutPad; ga.K = L.Cin; ga.nbatch = nbs; ga.act = 0; ga.tiles_m = ga.tiles_n = 0; ga.MP = wa.TP; ga.stagger = 0;
    RiderArgs rw;
    rw.in_u0 = rw.in_u1 = rw.out_u0 = rw.out_u1 = 0;
    if (ride_in) { rw.tin = *ride_in; rw.in_u1 = ride_in->Tc * ((ride_in->C + 63) / 64); }
    if (ride_out) { rw.tout = *ride_out; rw.out_u1 = ride_out->Tc * ((ride_out->Cout + 63) / 64); }
    prof_begin(n, 0, 2, 2.0 * nbs * wa.Tc * (double)L.Cin * L.Cout, s);
    gemm_dma_launch(ga, (ride_in || ride_out) ? &rw : nullptr, L.pers > 1 ? L.pers : 0, s);
    prof_end(n, s);
}

// The run of even-dilation convs with RIDERS (tdnet_opts.overlap bit 64): ONE stream, the two row-parity chains interleaved, every
// transform except the first and the last riding in the fifth wave of the other chain's GEMM workgroups.  The 36 GEMM batches of a chunk
// are two launches, 24 + 12 batches (for the 512-channel layers: 1536 and 768 tiles = 2 and 1 per resident workgroup), because the two
// transforms between a chain's consecutive GEMMs depend on each other through neighbouring tiles -- out(i) must be COMPLETE before
// in(i + 1) starts -- and so need two launches of the other chain to ride on:
//     GA(E,i) + out(O,i-1) | GB(E,i) + in(O,i) | GA(O,i) + out(E,i) | GB(O,i) + in(E,i+1) | ...
// Every dependency is the stream order of whole launches: no events, no flags, nothing to wait for inside a kernel.
static int run_parity_chains_riders(tdnet* n, PathLayers& L, int h, int w, hipStream_t s) {
    const int sb = n->seg_block, nblk = (int)L.blocks.size();
    {
        BlockLayers& B = L.blocks[sb];
        if (n->seg_conv == 1) TD_TRY(run_conv(n, B.c1, n->bx, h, w, nullptr, n->seg_t[sb], s));
        if (B.has_ds) TD_TRY(run_conv(n, B.ds, n->bx, h, w, nullptr, n->seg_r[sb], s));
    }
    struct Job { const ConvLayer* L; WinoArgs wa[2]; const ConvLayer* ds; const float* ds_in; float* ds_out; };
    std::vector<Job> jobs;
    float* Vw[2] = {n->wino_v, n->wino_v2};
    float* Mw[2] = {n->wino_m, n->wino_m2};
    for (int b = sb; b < nblk; ++b) {
        BlockLayers& B = L.blocks[b];
        const float* xin = b == sb ? n->bx : n->seg_x[b - 1];
        if (!(b == sb && n->seg_conv == 1)) {
            Job j; j.L = &B.c1; j.ds = nullptr; j.ds_in = nullptr; j.ds_out = nullptr;
            for (int c = 0; c < 2; ++c) { WinoChunk ck; ck.ny = 2; ck.cy = c; j.wa[c] = wino_chunk_args(B.c1, xin, h, w, nullptr, n->seg_t[b], ck, Vw[c], Mw[c]); }
            jobs.push_back(j);
        }
        Job j; j.L = &B.c2; j.ds = (B.has_ds && b > sb) ? &B.ds : nullptr; j.ds_in = xin; j.ds_out = n->seg_r[b];
        for (int c = 0; c < 2; ++c) {
            WinoChunk ck; ck.ny = 2; ck.cy = c;
            j.wa[c] = wino_chunk_args(B.c2, n->seg_t[b], h, w, B.has_ds ? n->seg_r[b] : xin, n->seg_x[b], ck, Vw[c], Mw[c]);
        }
        jobs.push_back(j);
    }
    const int nj = (int)jobs.size();
    for (auto& j : jobs)
        if (!j.L->gdma || !gemm_dma_supports(j.L->Cin, j.L->Cout, j.L->tile) || j.L->wino != 4) return td_fail("internal: rider schedule on a conv the LDS-DMA GEMM cannot run");
    wino_transform_alone(n, *jobs[0].L, jobs[0].wa[0], false, s);                    // in(E,0)
    for (int i = 0; i < nj; ++i) {
        Job& J = jobs[i];
        if (J.ds) for (int c = 0; c < 2; ++c) TD_TRY(run_ds_rows(n, *J.ds, J.ds_in, h, w, J.ds_out, 2, c, s));   // its input: out(.,i-2), long complete
        wino_gemm_with_rider(n, *J.L, J.wa[0], 0, 24, nullptr, i > 0 ? &jobs[i - 1].wa[1] : nullptr, s);          // GA(E,i) + out(O,i-1)
        wino_gemm_with_rider(n, *J.L, J.wa[0], 24, 12, &J.wa[1], nullptr, s);                                      // GB(E,i) + in(O,i)
        wino_gemm_with_rider(n, *J.L, J.wa[1], 0, 24, nullptr, &J.wa[0], s);                                       // GA(O,i) + out(E,i)
        wino_gemm_with_rider(n, *J.L, J.wa[1], 24, 12, i + 1 < nj ? &jobs[i + 1].wa[0] : nullptr, nullptr, s);     // GB(O,i) + in(E,i+1)
    }
    wino_transform_alone(n, *jobs[nj - 1].L, jobs[nj - 1].wa[1], true, s);           // out(O,last)
    return 0;
}

// The 1x1 stride-1 downsample conv (resnet.py:172-177) on the image rows y = ny * i + cy only: a batched GEMM, batch = row, M = W pixels,
// row pitch ny * W pixels, one weight set (td_gemm.h GemmArgs.wshare).
static int run_ds_rows(tdnet* n, const ConvLayer& L, const float* in, int H, int W, float* out, int ny, int cy, hipStream_t s) {
    if (L.KS != 1 || L.stride != 1 || L.h16 || !L.pers || !gemm_supports(L.Cin)) return td_fail("internal: downsample conv cannot run on image rows");
    const int rows = (H - cy + ny - 1) / ny;
    if (rows <= 0) return 0;
    GemmArgs ga;
    ga.a = in + (size_t)cy * W * L.Cin; ga.wp = L.d_wp; ga.bias = L.d_bias; ga.resid = nullptr; ga.out = out + (size_t)cy * W * L.Cout;
    ga.M = W; ga.N = L.Cout; ga.NPad = L.CoutPad; ga.K = L.Cin; ga.nbatch = rows; ga.act = L.act; ga.tiles_m = ga.tiles_n = 0; ga.MP = ny * W;
    ga.stagger = L.stagger; ga.wshare = 1;
    prof_begin(n, 0, 0, 2.0 * rows * W * (double)L.Cin * L.Cout, s);
    gemm_launch(ga, L.tile, L.pers > 1 ? L.pers : 0, s);
    prof_end(n, s);
    return 0;
}

// The trailing run of even-dilation Winograd convs as two row-parity chains (plan_chains): chain 0 on s, chain 1 on n->chain2.
// Each chain is an in-order sequence transform -> GEMMs -> transform -> ..., so nothing synchronises the two between fork and join;
// while one chain's GEMM holds the matrix pipes, the other's transforms (HBM-bound, one wave per SIMD beside the GEMM's three) run
// under it, and a chain's GEMM workgroups start as the other's retire.  Host enqueue order alternates between the chains so that
// neither stream runs dry while the other's launches are being issued.
static int run_parity_chains_riders(tdnet* n, PathLayers& L, int h, int w, hipStream_t s);

// The run on a PARTITIONED chip (tdnet_opts.cu_reserve, round 4).  Every earlier schedule asked the workgroup dispatcher to co-run an
// HBM-bound transform beside a persistent GEMM on the same CUs, which it does not do for a kernel that arrives later.
// Here the GEMMs of BOTH chains run back to back on part_g -- a queue that owns all but R CUs -- and the transforms of both chains on
// part_t, a queue that owns the other R: while G(O,i) holds the matrix pipes of its CUs, out(E,i) and in(E,i+1) stream through the
// reserved ones, so that G(E,i+1) finds its operand ready when G(O,i) retires:
//     part_t:  in(E,0) in(O,0) | out(E,0) in(E,1) | out(O,0) in(O,1) | out(E,1) in(E,2) | ...
//     part_g:          G(E,0)  |      G(O,0)      |      G(E,1)      |      G(O,1)      | ...
// ev_in[c][i]: part_g waits for it before G(c,i); ev_g[c][i]: part_t waits for it before out(c,i).  The host enqueues in a
// topological order of these dependencies (the emulator, which runs launches in issue order, checks exactly that).
static int run_parity_chains_partitioned(tdnet* n, PathLayers& L, int h, int w, hipStream_t s) {
    const int sb = n->seg_block, nblk = (int)L.blocks.size();
    hipStream_t G = n->part_g, T = n->part_t;
    {
        BlockLayers& B = L.blocks[sb];
        if (n->seg_conv == 1) TD_TRY(run_conv(n, B.c1, n->bx, h, w, nullptr, n->seg_t[sb], s));
        if (B.has_ds) TD_TRY(run_conv(n, B.ds, n->bx, h, w, nullptr, n->seg_r[sb], s));
    }
    struct Job { const ConvLayer* L; WinoArgs wa[2]; const ConvLayer* ds_after; const float* ds_in; float* ds_out; };
    std::vector<Job> jobs;
    float* Vw[2] = {n->wino_v, n->wino_v2};
    float* Mw[2] = {n->wino_m, n->wino_m2};
    for (int b = sb; b < nblk; ++b) {
        BlockLayers& B = L.blocks[b];
        const float* xin = b == sb ? n->bx : n->seg_x[b - 1];
        if (!(b == sb && n->seg_conv == 1)) {
            Job j; j.L = &B.c1;
            // the block's 1x1 downsample (rows of one parity, GemmArgs.wshare) follows conv1's GEMMs of that parity on part_g: its input
            // -- the previous block's output rows -- is complete (in(c, conv1) read them), its output is conv2's residual
            j.ds_after = (B.has_ds && b > sb) ? &B.ds : nullptr; j.ds_in = xin; j.ds_out = n->seg_r[b];
            for (int c = 0; c < 2; ++c) { WinoChunk ck; ck.ny = 2; ck.cy = c; j.wa[c] = wino_chunk_args(B.c1, xin, h, w, nullptr, n->seg_t[b], ck, Vw[c], Mw[c]); }
            jobs.push_back(j);
        }
        Job j; j.L = &B.c2; j.ds_after = nullptr; j.ds_in = nullptr; j.ds_out = nullptr;
        for (int c = 0; c < 2; ++c) {
            WinoChunk ck; ck.ny = 2; ck.cy = c;
            j.wa[c] = wino_chunk_args(B.c2, n->seg_t[b], h, w, B.has_ds ? n->seg_r[b] : xin, n->seg_x[b], ck, Vw[c], Mw[c]);
        }
        jobs.push_back(j);
    }
    const int nj = (int)jobs.size();
    if ((size_t)nj + 1 > n->ev_in[0].size()) return td_fail("internal: partitioned run has more convs than events");
    for (auto& j : jobs) if (j.L->wino != 4 || !j.L->vw) return td_fail("internal: partitioned run on a conv without chunked F(4x4) transforms");
    TD_HIP(hipEventRecord(n->ev_cfork, s));
    TD_HIP(hipStreamWaitEvent(T, n->ev_cfork, 0));
    TD_HIP(hipStreamWaitEvent(G, n->ev_cfork, 0));
    auto gemms = [&](const Job& J, int c) {
        const ConvLayer& C = *J.L;
        const WinoArgs& wa = J.wa[c];
        GemmArgs ga;
        ga.a = wa.V; ga.wp = C.d_wp; ga.bias = C.d_zero; ga.resid = nullptr; ga.out = const_cast<float*>(wa.Mb);
        ga.M = wa.Tc; ga.N = C.Cout; ga.NPad = C.CoutPad; ga.K = C.Cin; ga.nbatch = 36; ga.act = 0; ga.tiles_m = ga.tiles_n = 0; ga.MP = wa.TP; ga.stagger = 0;
        prof_begin(n, 0, 2, 2.0 * 36 * wa.Tc * (double)C.Cin * C.Cout, G);
        const int cap = C.pers > 1 ? C.pers : n->part_grid;
        if (C.gdma && gemm_dma_supports(C.Cin, C.Cout, C.tile)) gemm_dma_launch(ga, nullptr, cap, G);
        else gemm_launch(ga, C.tile, C.pers > 1 ? C.pers : n->part_grid * gemm_blocks_per_cu(C.tile) / 3, G);
        prof_end(n, G);
    };
    for (int c = 0; c < 2; ++c) {
        wino_transform_alone(n, *jobs[0].L, jobs[0].wa[c], false, T);
        TD_HIP(hipEventRecord(n->ev_in[c][0], T));
    }
    for (int i = 0; i < nj; ++i) {
        const Job& J = jobs[i];
        for (int c = 0; c < 2; ++c) {
            TD_HIP(hipStreamWaitEvent(G, n->ev_in[c][i], 0));
            gemms(J, c);
            TD_HIP(hipEventRecord(n->ev_g[c][i], G));
            if (J.ds_after) TD_TRY(run_ds_rows(n, *J.ds_after, J.ds_in, h, w, J.ds_out, 2, c, G));
        }
        for (int c = 0; c < 2; ++c) {
            TD_HIP(hipStreamWaitEvent(T, n->ev_g[c][i], 0));
            wino_transform_alone(n, *J.L, J.wa[c], true, T);
            if (i + 1 < nj) {
                wino_transform_alone(n, *jobs[i + 1].L, jobs[i + 1].wa[c], false, T);
                TD_HIP(hipEventRecord(n->ev_in[c][i + 1], T));
            }
        }
    }
    // part_t ends with out(O, last); everything on part_g precedes it through ev_g -- except a trailing downsample, which the last job has not
    TD_HIP(hipEventRecord(n->ev_cjoin, T));
    TD_HIP(hipStreamWaitEvent(s, n->ev_cjoin, 0));
    return 0;
}

static int run_parity_chains(tdnet* n, PathLayers& L, int h, int w, hipStream_t s) {
    if ((n->opts.overlap & 64) && (n->opts.overlap & 8)) return run_parity_chains_riders(n, L, h, w, s);
    if (n->part_g) return run_parity_chains_partitioned(n, L, h, w, s);
    const int sb = n->seg_block, nblk = (int)L.blocks.size();
    hipStream_t st[2] = {s, n->chain2};
    float* Vw[2] = {n->wino_v, n->wino_v2};
    float* Mw[2] = {n->wino_m, n->wino_m2};
    {   // the part of the first block that precedes the run: its downsample, and conv1 when the run starts at conv2
        BlockLayers& B = L.blocks[sb];
        if (n->seg_conv == 1) TD_TRY(run_conv(n, B.c1, n->bx, h, w, nullptr, n->seg_t[sb], s));
        if (B.has_ds) TD_TRY(run_conv(n, B.ds, n->bx, h, w, nullptr, n->seg_r[sb], s));
    }
    TD_HIP(hipEventRecord(n->ev_cfork, s));
    TD_HIP(hipStreamWaitEvent(n->chain2, n->ev_cfork, 0));
    // overlap bit 4: chain 1 starts when chain 0's FIRST input transform is done, i.e. together with chain 0's first GEMM.  Started
    // together the chains run in lockstep (transform beside transform, GEMM beside GEMM: profiles/r03a_timeline_*); half a conv apart,
    // one chain's transforms meet the other's GEMMs.
    bool stagger_pending = (n->opts.overlap & 4) != 0;
    for (int b = sb; b < nblk; ++b) {
        BlockLayers& B = L.blocks[b];
        const float* xin = b == sb ? n->bx : n->seg_x[b - 1];
        for (int c = 0; c < 2; ++c) {
            WinoChunk ck; ck.ny = 2; ck.cy = c;
            if (!(b == sb && n->seg_conv == 1)) TD_TRY(run_wino(n, B.c1, xin, h, w, nullptr, n->seg_t[b], st[c], nullptr, ck, Vw[c], Mw[c], c == 0 && stagger_pending ? n->ev_cstag : nullptr, c == 0 && stagger_pending ? n->chain2 : nullptr));
            if (c == 0 && !(b == sb && n->seg_conv == 1)) stagger_pending = false;
            if (B.has_ds && b > sb) TD_TRY(run_ds_rows(n, B.ds, xin, h, w, n->seg_r[b], 2, c, st[c]));
        }
        for (int c = 0; c < 2; ++c) {
            WinoChunk ck; ck.ny = 2; ck.cy = c;
            TD_TRY(run_wino(n, B.c2, n->seg_t[b], h, w, B.has_ds ? n->seg_r[b] : xin, n->seg_x[b], st[c], nullptr, ck, Vw[c], Mw[c], c == 0 && stagger_pending ? n->ev_cstag : nullptr, c == 0 && stagger_pending ? n->chain2 : nullptr));
            if (c == 0) stagger_pending = false;
        }
    }
    TD_HIP(hipEventRecord(n->ev_cjoin, n->chain2));
    TD_HIP(hipStreamWaitEvent(s, n->ev_cjoin, 0));
    return 0;
}

static int encode_frame(tdnet* n, PathLayers& L, const float* img, hipStream_t s) {
    const int DV = n->DV;
    // backbone (resnet.py:204-215)
    run_stem_pre(n, img, n->H, n->W, n->img4, s, n->opts.fusion);
    if (n->deep) {                                                     // resnet.py:122-131
        TD_TRY(run_conv(n, L.stem, n->img4, n->H, n->W, nullptr, n->s1b, s));
        TD_TRY(run_conv(n, L.stem2, n->s1b, n->H1, n->W1, nullptr, n->s1, s));
        TD_TRY(run_conv(n, L.stem3, n->s1, n->H1, n->W1, nullptr, n->br, s));   // 128 ch at H1 x W1 -> br (sized for it below)
    } else {
        TD_TRY(run_conv(n, L.stem, n->img4, n->H, n->W, nullptr, n->s1, s));
    }
    run_maxpool(n, n->deep ? n->br : n->s1, n->H1, n->W1, n->SC, n->bx, s, n->opts.fusion, n->act16 ? ((n->deep || L.stem.out16) ? 2 : 1) : 0);
    int ch = n->H2, cw = n->W2;
    for (size_t bi = 0; bi < L.blocks.size(); ++bi) {
        BlockLayers& B = L.blocks[bi];
        if ((int)bi == n->seg_block) {                                 // the rest of the backbone as two row-parity chains
            if (ch != n->h || cw != n->w) return td_fail("internal: the chained run is not at the output resolution");
            TD_TRY(run_parity_chains(n, L, ch, cw, s));
            break;
        }
        int oh, ow;
        // fp16-activation mode: the LAST conv of the backbone writes fp32 while its residual is an fp16 map -- not in place (4-byte
        // stores over 2-byte elements other lanes still have to read): it goes to br, free here (the last block has no downsample)
        const bool last16 = n->act16 && &B == &L.blocks.back();
        if (B.bott) {                                                  // resnet.py:91-111
            TD_TRY(run_conv(n, B.c1, n->bx, ch, cw, nullptr, n->bt, s));                        // 1x1, input resolution
            TD_TRY(run_conv(n, B.c2, n->bt, ch, cw, nullptr, n->bu, s, &oh, &ow));              // 3x3 (stride, dilation)
            const float* res = n->bx;
            if (B.has_ds) { TD_TRY(run_conv(n, B.ds, n->bx, ch, cw, nullptr, n->br, s)); res = n->br; }
            TD_TRY(run_conv(n, B.c3, n->bu, oh, ow, res, last16 ? n->br : n->bx, s));         // 1x1 x4 + residual + ReLU (in place when res == bx)
        } else {
            TD_TRY(run_conv(n, B.c1, n->bx, ch, cw, nullptr, n->bt, s, &oh, &ow));
            const float* res = n->bx;
            if (B.has_ds) { TD_TRY(run_conv(n, B.ds, n->bx, ch, cw, nullptr, n->br, s)); res = n->br; }
            TD_TRY(run_conv(n, B.c2, n->bt, oh, ow, res, last16 ? n->br : n->bx, s));   // in-place on bx when res == bx (same element)
        }
        ch = oh; cw = ow;
    }
    float* c4 = n->c4 = n->seg_block >= 0 ? n->seg_x.back() : n->act16 ? n->br : n->bx;
    // pyramid pooling slice (td4_psp18.py:271-284)
    if (n->cfg.model == 1) {                                           // pspnet.py:73-89: PSPHead on c4, no temporal state
        run_ppm(n, c4, n->h, n->w, n->C, n->C, n->C / 4, L.d_ppm_w, L.d_ppm_b, 0, n->rowpart, n->pooled, n->ppmfeat, n->z, s);
        TD_TRY(run_conv(n, L.head3, n->z, n->h, n->w, nullptr, n->headmid, s));
        TD_TRY(run_classifier(n, n->headmid, n->Lq, n->MID, n->cfg.nclass, L.d_cls_w, L.d_cls_b, n->lowres, s));
        return n->failed ? -1 : 0;
    }
    run_ppm(n, c4, n->h, n->w, n->C, n->C / 2, n->C / 8, L.d_ppm_w, L.d_ppm_b, L.pid, n->rowpart, n->pooled, n->ppmfeat, n->z, s);
    // Encoding, pre=False (transformer.py:52-56) and pre=True (:34-50) -> pending cache entry; q_ and v_ are the stride-4 subsample of
    // q_cur / v_cur.  The q / k branches (512 -> 64 -> 64; the k branch on the 16x smaller key grid: 16 workgroups) are short,
    // latency-bound launches that depend only on z: with fusion bit 1 they run on the side stream beside the w_vs GEMM.
    const int slot = free_slot(n);
    if (slot < 0) return td_fail("internal: no free cache slot");
    CacheSlot& cs = n->slots[slot];
    const bool beside = (n->opts.fusion & 1) != 0;
    hipStream_t qs = beside ? n->side : s;
    if (beside) {
        TD_HIP(hipEventRecord(n->ev_fork2, s));
        TD_HIP(hipStreamWaitEvent(qs, n->ev_fork2, 0));
    }
    if (!beside) TD_TRY(run_conv(n, L.enc_v, n->z, n->h, n->w, nullptr, n->v_cur, s));
    TD_TRY(run_conv(n, L.enc_q0, n->z, n->h, n->w, nullptr, n->q1, qs));
    TD_TRY(run_conv(n, L.enc_q1, n->q1, n->h, n->w, nullptr, n->q_cur, qs));
    TD_TRY(run_conv(n, L.enc_k0, n->z, n->h, n->w, nullptr, n->k1, qs));
    TD_TRY(run_conv(n, L.enc_k1, n->k1, n->hk, n->wk, nullptr, cs.k, qs));
    prof_begin(n, 2, false, 0, qs);
    TD_LAUNCH(k_subsample, dim3(td_grid_for((long)n->Lk * 16)), dim3(256), 0, qs, (const float*)n->q_cur, cs.q, n->w, 64, n->hk, n->wk, 4);
    prof_end(n, qs);
    if (beside) {
        TD_HIP(hipEventRecord(n->ev_join2, qs));
        TD_TRY(run_conv(n, L.enc_v, n->z, n->h, n->w, nullptr, n->v_cur, s));
    }
    prof_begin(n, 2, false, 0, s);
    TD_LAUNCH(k_subsample, dim3(td_grid_for((long)n->Lk * (DV / 4))), dim3(256), 0, s, (const float*)n->v_cur, cs.v, n->w, DV, n->hk, n->wk, 4);
    prof_end(n, s);
    if (beside) TD_HIP(hipStreamWaitEvent(s, n->ev_join2, 0));
    n->pending_slot = slot;
    return n->failed ? -1 : 0;
}

// chain_launched: launch_chain() already ran for this frame (it read the FIFO as it is now)
static int finish_frame(tdnet* n, PathLayers& L, bool steady, hipStream_t s, int prelaunch_pos = -1) {
    const int DV = n->DV;
    const float* feat = n->v_cur;
    int stats_nstr = 0;
    if (steady) {
        TD_HIP(hipStreamWaitEvent(s, n->ev_join, 0));                   // join: v' of the newest cached frame is ready
        const CacheSlot& ck = n->slots[n->fifo[n->FIFO - 1]];
        const AtnLayer& A = L.atn[n->P == 4 ? 2 : 0];                   // td4_psp18.py:147 / td2_psp50.py:120
        stats_nstr = (n->opts.fusion & 2) ? attn_strips(n->Lq, DV) : 0;                                         // LayerNorm strip statistics from the epilogue
        if (run_attention(n, n->q_cur, ck.k, n->vp_read, A.d_bias, n->v_cur, n->Lq, n->Lk, DV, n->feat, s, n->opts.attention,
                          stats_nstr ? n->ln_part : nullptr)) return -1;                                                  // v4 + v_cur
        feat = n->feat;
    } else {
        // warm-up (td4_psp18.py:142-143): feat = v_cur; keep a copy so the "feat" stage is well defined
        TD_HIP(hipMemcpyAsync(n->feat, n->v_cur, (size_t)n->Lq * DV * sizeof(float), hipMemcpyDeviceToDevice, s));
        feat = n->feat;
    }
    // FIFO push (td4_psp18.py:153-154, :123-134): host bookkeeping only -- the entry's data was written by encode_frame.  It is the
    // LAST thing a frame does (a frame whose head fails to launch is not in the FIFO), except with overlap bit 128, whose pre-launched
    // chain needs the FIFO as the next frame will see it: there the commit precedes the head.
    auto commit = [&]() {
        if (n->pending_slot < 0) return;
        const int slot = n->pending_slot;
        n->pending_slot = -1;
        fifo_commit(n, slot);
    };
    const bool prelaunch = prelaunch_pos >= 0 && (n->opts.overlap & 128) && n->vp2;
    if (prelaunch) commit();
    // overlap bit 128: the NEXT frame's cache-only chain starts here, when this frame's final attention is done -- beside the HBM-bound
    // rest of this frame (matrix pipes idle) instead of beside the next frame's stem and layer1, which it slowed by 20-50 %.  It assumes
    // the next call is pos_id + 1 on an untouched FIFO; forward_lowres checks and falls back to launching the chain itself.
    n->pre_valid = false;
    if (prelaunch && (int)n->fifo.size() >= n->FIFO) {
        float* target = n->vp_read == n->vp ? n->vp2 : n->vp;
        if (launch_chain(n, n->paths[prelaunch_pos], s, target, n->fifo[0], n->P == 4 ? n->fifo[1] : -1, n->P == 4 ? n->fifo[2] : -1)) return -1;
        n->pre_valid = true; n->pre_pos = prelaunch_pos; n->pre_epoch = n->fifo_epoch; n->pre_vp = target;
    }
    // fusion bit 4: the normalised map is never written -- the head's Winograd input transform normalises while it reads `feat`
    const bool ln_in_head = (n->opts.fusion & 4) && L.head3.wino;
    const bool ln16 = L.head3.in16;                                     // fp16 mode: n->ln holds the map as fp16; the fp32 stage is made on request
    run_layernorm(n, feat, n->Lq, DV, L.d_ln_g, L.d_ln_b, n->ln_part, n->ln_mean, n->ln_rstd, ln_in_head ? nullptr : n->ln, s, stats_nstr, ln16);
    n->ln_pending = ln_in_head || ln16;
    n->ln_path = (int)(&L - &n->paths[0]);
    if (ln_in_head) {
        const LnFuse lf = {n->ln_mean, n->ln_rstd, L.d_ln_g, L.d_ln_b};
        TD_TRY(run_conv(n, L.head3, feat, n->h, n->w, nullptr, n->headmid, s, nullptr, nullptr, &lf));
    } else
    TD_TRY(run_conv(n, L.head3, n->ln, n->h, n->w, nullptr, n->headmid, s));
    TD_TRY(run_classifier(n, n->headmid, n->Lq, n->MID, n->cfg.nclass, L.d_cls_w, L.d_cls_b, n->lowres, s));
    if (n->failed) return -1;
    commit();
    return 0;
}

// overlap bit 128: a pre-launched chain that will never be used (reset, an external cache push) may still be READING cache slots on the
// side stream; whoever writes slots next -- the next frame's Encoding, a pushed entry -- first waits for it on its own stream.
static int retire_stale_chain(tdnet* n, hipStream_t s) {
    if (!n->chain_stale) return 0;
    n->chain_stale = false;
    TD_HIP(hipStreamWaitEvent(s, n->ev_join, 0));
    return 0;
}

static int frame_checks(tdnet* n, int pos_id, const char* who) {
    if (!n->finalized) return td_fail("%s: weights not finalized (the HIP path never runs on random init)", who);
    if (pos_id < 0 || pos_id >= n->P) return td_fail("%s: pos_id %d out of range 0..%d", who, pos_id, n->P - 1);
    return 0;
}

// Error path of a frame: whatever the internal streams (cache-only attention chain, second row-parity chain) were given before the
// failure is joined back into the caller's stream, so that a failed call leaves no work of this handle running unordered behind it.
static void rejoin_streams(tdnet* n, hipStream_t s) {
    for (hipStream_t c : {n->side, n->chain2, n->part_g, n->part_t}) {
        if (!c) continue;
        hipEvent_t& e = c == n->side ? n->ev_join : n->ev_cjoin;
        if (e && hipEventRecord(e, c) == hipSuccess) (void)hipStreamWaitEvent(s, e, 0);
    }
}

// ---- the second chain's stream must really be a second QUEUE --------------------------------------------------------------------------------
// HIP deals a process's streams onto a small pool of hardware queues per priority class (4 by default), reusing queues once the pool is
// full; two streams on one queue run their kernels one after the other.  With two or more other normal-priority streams alive in the
// process, `chain2` used to land on the CALLER's queue: the two row-parity chains serialised and the headline fell from 275 to 183 frames/s
// (tools/ab_opts.py under TDNET_PROBE_EXTRA_STREAMS, profiles/r04k_headline_vs_extra_streams_in_the_process.txt).  So the first frame on a
// given caller stream checks: two 40-us spin kernels, one on the caller's stream and one on chain2, started together -- ~40 us for the
// pair = two queues, ~80 us = one.  If they serialise, chain2 is replaced by a fresh stream (the rejected ones stay alive until the handle
// dies, or the pool would hand the same queue out again), at most six times.  One host synchronisation per attempt, once per handle and stream.
// Measured (profiles/r04k_*): one busy handle beside 0 / 1 / 2 / 3 idle ones 335 / 212 / 335 / 212 frames/s before, 333 / 334 / 333 / 334
// with the check; two extra streams in the process 193-275 -> 273.  NOT cured: three or more extra normal-priority streams created before
// the handle's own (182 frames/s although the spin pair runs side by side) -- something below HIP's queue pool that a marker kernel beside an
// oversubscribed grid could not tell apart from ordinary occupancy (tried, removed).
#ifndef TD_EMU
__global__ void k_queue_probe_spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
static int streams_share_a_queue(hipStream_t a, hipStream_t x, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, bool* shared) {
    float worst = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {                                // the better of two: a context switch on the host must not look like a shared queue
        TD_HIP(hipEventRecord(e0, a));
        TD_HIP(hipStreamWaitEvent(x, e0, 0));
        TD_LAUNCH(k_queue_probe_spin, dim3(1), dim3(64), 0, a, 4000ull);
        TD_LAUNCH(k_queue_probe_spin, dim3(1), dim3(64), 0, x, 4000ull);
        TD_HIP(hipEventRecord(e1, a));
        TD_HIP(hipEventRecord(e2, x));
        TD_HIP(hipStreamWaitEvent(a, e2, 0));                          // the caller's stream stays ordered behind everything this enqueued
        TD_HIP(hipEventSynchronize(e1));
        TD_HIP(hipEventSynchronize(e2));
        float t1 = 0.f, t2 = 0.f;
        TD_HIP(hipEventElapsedTime(&t1, e0, e1));
        TD_HIP(hipEventElapsedTime(&t2, e0, e2));
        worst = std::min(worst, std::max(t1, t2));
    }
    *shared = worst > 0.064f;                                          // 40 us each: 40-45 us side by side, 80+ us one after the other
    if (getenv("TDNET_QUEUE_CHECK_VERBOSE")) fprintf(stderr, "tdnet queue check: spin pair %.1f us\n", worst * 1e3f);
    return 0;
}
#endif
static int place_chain_stream(tdnet* n, hipStream_t s) {
    if (!n->chain2 || n->part_g || (n->placed && n->placed_for == (void*)s)) return 0;
    n->placed = true; n->placed_for = (void*)s;
#ifndef TD_EMU
    if (getenv("TDNET_NO_QUEUE_CHECK")) return 0;                      // A/B of this very mechanism (tools/ab_opts.py)
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    TD_HIP(hipEventCreate(&e0)); TD_HIP(hipEventCreate(&e1)); TD_HIP(hipEventCreate(&e2));
    int rc = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {
        bool shared = false;
        if ((rc = streams_share_a_queue(s, n->chain2, e0, e1, e2, &shared)) != 0 || !shared) break;
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
        n->retired_streams.push_back(n->chain2);
        n->chain2 = fresh;
        n->chain_replaced++;
    }
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    return rc;
#else
    return 0;
#endif
}

static int forward_lowres_impl(tdnet* n, const float* img, int pos_id, hipStream_t s);
static int forward_lowres(tdnet* n, const float* img, int pos_id, hipStream_t s) {
    const int rc = forward_lowres_impl(n, img, pos_id, s);
    if (rc && n && n->finalized) { rejoin_streams(n, s); n->pending_slot = -1; }   // the failed frame is dropped: it never reaches the FIFO
    return rc;
}
static int forward_lowres_impl(tdnet* n, const float* img, int pos_id, hipStream_t s) {
    if (frame_checks(n, pos_id, "tdnet_forward")) return -1;
    if (n->pending_slot >= 0) return td_fail("tdnet_forward: a frame encoded with tdnet_encode is waiting for tdnet_propagate");
    PathLayers& L = n->paths[pos_id];
    n->nrec = 0;
    n->failed = false;
    TD_TRY(place_chain_stream(n, s));
    TD_TRY(retire_stale_chain(n, s));
    const bool steady = n->cfg.model != 1 && (int)n->fifo.size() >= n->FIFO;
    if (steady) {
        if (n->pre_valid && n->pre_pos == pos_id && n->pre_epoch == n->fifo_epoch) n->vp_read = n->pre_vp;   // launched at the end of the previous frame
        else if (launch_chain_now(n, L, s)) return -1;                 // overlaps the backbone below
    }
    n->pre_valid = false;
    if (encode_frame(n, L, img, s)) return -1;
    if (n->cfg.model == 1) return 0;
    return finish_frame(n, L, steady, s, (pos_id + 1) % n->P);
}

extern "C" int tdnet_forward(tdnet_t* n, const float* img, int pos_id, float* logits, void* stream) {
    if (!n || !img || !logits) return td_fail("tdnet_forward: null argument");
    TD_ON_DEVICE(n, -1);
    hipStream_t s = (hipStream_t)stream;
    if (forward_lowres(n, img, pos_id, s)) return -1;
    prof_begin(n, 2, false, 0, s);
    launch_upsample(n->lowres, n->cfg.nclass, n->h, n->w, n->H, n->W, logits, s);
    prof_end(n, s);
    TD_HIP(hipGetLastError());
    return 0;
}
extern "C" int tdnet_argmax(tdnet_t* n, const float* logits, int32_t* labels, void* stream) {
    if (!n || !logits || !labels) return td_fail("tdnet_argmax: null argument");
    TD_ON_DEVICE(n, -1);
    TD_LAUNCH(k_argmax, dim3(td_grid_for((long)n->H * n->W)), dim3(256), 0, (hipStream_t)stream, logits, labels, n->cfg.nclass, (long)n->H * n->W);
    TD_HIP(hipGetLastError());
    return 0;
}
extern "C" int tdnet_forward_labels(tdnet_t* n, const float* img, int pos_id, int32_t* labels, void* stream) {
    if (!n || !img || !labels) return td_fail("tdnet_forward_labels: null argument");
    TD_ON_DEVICE(n, -1);
    hipStream_t s = (hipStream_t)stream;
    if (forward_lowres(n, img, pos_id, s)) return -1;
    prof_begin(n, 2, false, 0, s);
    TD_LAUNCH(k_upsample_argmax, dim3(td_grid_for((long)n->H * n->W)), dim3(256), 0, s, (const float*)n->lowres, labels, n->cfg.nclass,
              n->h, n->w, n->H, n->W);
    prof_end(n, s);
    TD_HIP(hipGetLastError());
    return 0;
}
// ---- split frame + cache transport (path-parallel single stream) ------------------------------------
// Rank g of a path-parallel group serves the frames t = g (mod W): it encodes its frame as soon as the image is there, publishes
// the resulting cache entry, receives the entries of the frames in between from its peers (in frame order) and only then
// propagates.  The FIFO of every rank therefore holds exactly what the sequential td4_psp18.py:123-154 would hold.
extern "C" int tdnet_encode(tdnet_t* n, const float* img, int pos_id, void* stream) {
    if (!n || !img) return td_fail("tdnet_encode: null argument");
    TD_ON_DEVICE(n, -1);
    if (frame_checks(n, pos_id, "tdnet_encode")) return -1;
    if (n->cfg.model == 1) return td_fail("tdnet_encode: pspnet has no temporal state; use tdnet_forward");
    if (n->pending_slot >= 0) return td_fail("tdnet_encode: the previous encoded frame has not been propagated");
    n->nrec = 0;
    n->failed = false;
    TD_TRY(place_chain_stream(n, (hipStream_t)stream));
    TD_TRY(retire_stale_chain(n, (hipStream_t)stream));
    if (encode_frame(n, n->paths[pos_id], img, (hipStream_t)stream)) { rejoin_streams(n, (hipStream_t)stream); return -1; }
    n->pending_pos = pos_id;
    TD_HIP(hipGetLastError());
    return 0;
}
static int propagate_lowres(tdnet* n, hipStream_t s) {
    if (n->pending_slot < 0) return td_fail("tdnet_propagate: no encoded frame (call tdnet_encode first)");
    PathLayers& L = n->paths[n->pending_pos];
    const bool steady = (int)n->fifo.size() >= n->FIFO;
    n->pre_valid = false;
    if ((steady && launch_chain_now(n, L, s)) || finish_frame(n, L, steady, s)) { rejoin_streams(n, s); n->pending_slot = n->pending_pos = -1; return -1; }
    return 0;
}
extern "C" int tdnet_propagate(tdnet_t* n, float* logits, void* stream) {
    if (!n || !logits) return td_fail("tdnet_propagate: null argument");
    TD_ON_DEVICE(n, -1);
    hipStream_t s = (hipStream_t)stream;
    if (propagate_lowres(n, s)) return -1;
    launch_upsample(n->lowres, n->cfg.nclass, n->h, n->w, n->H, n->W, logits, s);
    TD_HIP(hipGetLastError());
    return 0;
}
extern "C" int tdnet_propagate_labels(tdnet_t* n, int32_t* labels, void* stream) {
    if (!n || !labels) return td_fail("tdnet_propagate_labels: null argument");
    TD_ON_DEVICE(n, -1);
    hipStream_t s = (hipStream_t)stream;
    if (propagate_lowres(n, s)) return -1;
    TD_LAUNCH(k_upsample_argmax, dim3(td_grid_for((long)n->H * n->W)), dim3(256), 0, s, (const float*)n->lowres, labels, n->cfg.nclass,
              n->h, n->w, n->H, n->W);
    TD_HIP(hipGetLastError());
    return 0;
}
extern "C" int tdnet_cache_dims(const tdnet_t* n, int* Lk, int* dk, int* dv) {
    if (!n) return td_fail("tdnet_cache_dims: null handle");
    if (n->cfg.model == 1) return td_fail("tdnet_cache_dims: pspnet has no cache");
    if (Lk) *Lk = n->Lk;
    if (dk) *dk = 64;
    if (dv) *dv = n->DV;
    return 0;
}
extern "C" int tdnet_cache_export(tdnet_t* n, float* q, float* k, float* v, void* stream) {
    if (!n || !q || !k || !v) return td_fail("tdnet_cache_export: null argument");
    if (n->pending_slot < 0) return td_fail("tdnet_cache_export: no encoded frame (call tdnet_encode first)");
    TD_ON_DEVICE(n, -1);
    const CacheSlot& c = n->slots[n->pending_slot];
    hipStream_t s = (hipStream_t)stream;
    TD_HIP(hipMemcpyAsync(q, c.q, (size_t)n->Lk * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    TD_HIP(hipMemcpyAsync(k, c.k, (size_t)n->Lk * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    TD_HIP(hipMemcpyAsync(v, c.v, (size_t)n->Lk * n->DV * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
extern "C" int tdnet_cache_push(tdnet_t* n, const float* q, const float* k, const float* v, void* stream) {
    if (!n || !q || !k || !v) return td_fail("tdnet_cache_push: null argument");
    if (n->cfg.model == 1) return td_fail("tdnet_cache_push: pspnet has no cache");
    TD_ON_DEVICE(n, -1);
    const int slot = free_slot(n);
    if (slot < 0) return td_fail("internal: no free cache slot");
    const CacheSlot& c = n->slots[slot];
    hipStream_t s = (hipStream_t)stream;
    if (n->pre_valid) { n->chain_stale = true; n->pre_valid = false; }  // the FIFO changes behind a pre-launched chain's back
    TD_TRY(retire_stale_chain(n, s));
    TD_HIP(hipMemcpyAsync(c.q, q, (size_t)n->Lk * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    TD_HIP(hipMemcpyAsync(c.k, k, (size_t)n->Lk * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    TD_HIP(hipMemcpyAsync(c.v, v, (size_t)n->Lk * n->DV * sizeof(float), hipMemcpyDeviceToDevice, s));
    fifo_commit(n, slot);
    n->fifo_epoch++;                                                   // the FIFO changed behind a pre-launched chain's back
    return 0;
}
extern "C" int tdnet_reset(tdnet_t* n) {
    if (!n) return td_fail("tdnet_reset: null handle");
    n->fifo_epoch++;
    if (n->pre_valid) n->chain_stale = true;                           // it may still be reading slots: the next writer waits (retire_stale_chain)
    n->pre_valid = false;
    n->fifo.clear();
    n->last_slot = -1;
    n->pending_slot = n->pending_pos = -1;
    return 0;
}
extern "C" int tdnet_fifo_len(const tdnet_t* n) { return n ? (int)n->fifo.size() : -1; }

// ---------------------------------------------------------------------------------------------------------------
// introspection
// ---------------------------------------------------------------------------------------------------------------
extern "C" long tdnet_get_stage(tdnet_t* n, const char* name, float* host, size_t capacity) {
    if (!n || !name || !host) return td_fail("tdnet_get_stage: null argument");
    TD_ON_DEVICE(n, -1);
    const std::string s = name;
    const float* src = nullptr;
    long rows = n->Lq, C = 0;
    bool nhwc_map = true, planar = false;
    if (s == "c4") { src = n->c4 ? n->c4 : n->bx; C = n->C; }
    else if (s == "z") { src = n->z; C = n->cfg.model == 1 ? 2 * n->C : n->C; }
    else if (s == "lowres") { src = n->lowres; C = n->cfg.nclass; planar = true; }
    else if (n->cfg.model == 1) return td_fail("tdnet_get_stage: stage \"%s\" does not exist in the single-frame PSPNet", name);
    else if (s == "v_cur") { src = n->v_cur; C = n->DV; }
    else if (s == "feat") { src = n->feat; C = n->DV; }
    else if (s == "ln") { src = n->ln; C = n->DV; }
    else if (s == "q_cur") { src = n->q_cur; C = 64; nhwc_map = false; }
    else if (s == "cache_q" || s == "cache_k" || s == "cache_v") {
        if (n->last_slot < 0) return td_fail("tdnet_get_stage: no frame cached yet");
        const CacheSlot& c = n->slots[n->last_slot];
        src = s == "cache_q" ? c.q : s == "cache_k" ? c.k : c.v;
        C = s == "cache_v" ? n->DV : 64; rows = n->Lk; nhwc_map = false;
    } else return td_fail("tdnet_get_stage: unknown stage \"%s\"", name);
    const size_t count = (size_t)rows * C;
    if (capacity < count) return td_fail("tdnet_get_stage: capacity %zu < %zu", capacity, count);
    TD_HIP(hipDeviceSynchronize());
    if (s == "ln" && n->ln_pending) {                                  // fusion bit 4 skipped this map: materialise it now, same arithmetic
        const PathLayers& PL = n->paths[n->ln_path];
        TD_LAUNCH(k_ln_apply, dim3(td_grid_for((long)n->Lq * (n->DV / 4))), dim3(256), 0, (hipStream_t)0, (const float*)n->feat,
                  (const float*)n->ln_mean, (const float*)n->ln_rstd, (const float*)PL.d_ln_g, (const float*)PL.d_ln_b, n->ln, n->Lq, n->DV);
        TD_HIP(hipDeviceSynchronize());
        n->ln_pending = false;
    }
    if (nhwc_map && !planar) {                                         // [HW][C] -> [C][HW] like the reference's NCHW maps
        TD_LAUNCH(k_nhwc_to_nchw, dim3(td_grid_for((long)count)), dim3(256), 0, (hipStream_t)0, src, n->stage_tmp, rows, (int)C);
        TD_HIP(hipDeviceSynchronize());
        TD_HIP(hipMemcpy(host, n->stage_tmp, count * sizeof(float), hipMemcpyDeviceToHost));
    } else {
        TD_HIP(hipMemcpy(host, src, count * sizeof(float), hipMemcpyDeviceToHost));
    }
    return (long)count;
}

// Algorithmic FLOP of a steady-state frame, counted as the reference executes it (fc on Lq rows)
static double frame_flops(const tdnet* n) {
    const PathLayers& L = n->paths[0];
    double f = L.stem.flops_per_pixel() * n->H1 * n->W1;
    if (n->deep) f += (L.stem2.flops_per_pixel() + L.stem3.flops_per_pixel()) * n->H1 * n->W1;
    int ch = n->H2, cw = n->W2;
    for (size_t i = 0; i < L.blocks.size(); ++i) {
        const BlockSpec& s = n->bspec[i];
        const int oh = out_size(ch, 3, s.stride, s.dil1, s.dil1), ow = out_size(cw, 3, s.stride, s.dil1, s.dil1);
        const double M = (double)oh * ow;
        if (s.bott) f += (double)ch * cw * L.blocks[i].c1.flops_per_pixel() + M * (L.blocks[i].c2.flops_per_pixel() + L.blocks[i].c3.flops_per_pixel());
        else f += M * (L.blocks[i].c1.flops_per_pixel() + L.blocks[i].c2.flops_per_pixel());
        if (s.ds) f += M * L.blocks[i].ds.flops_per_pixel();
        ch = oh; cw = ow;
    }
    const double C = n->C;
    if (n->cfg.model == 1)
        return f + 2.0 * 50 * (C / 4) * C * 4 / 4 + (double)n->Lq * (2.0 * 2 * C * 9 * n->MID + 2.0 * n->MID * n->cfg.nclass);
    const double Lq = n->Lq, Lk = n->Lk, DV = n->DV;
    f += 2.0 * (1 + 4 + 9 + 36) * (C / 4) * C;                                          // pyramid 1x1 convs on the 50 bins
    f += Lq * (2.0 * C * DV + 2.0 * C * 64 + 2.0 * 64 * 64);                           // enc pre=False
    f += Lk * (2.0 * C * DV + 2 * (2.0 * C * 64 + 2.0 * 64 * 64));                     // enc pre=True (q_, k_, v_)
    if (n->P == 4) {
        f += 2 * (2.0 * Lk * Lk * (64 + DV) + 2.0 * Lk * DV * DV);                         // two cached-frame attentions + fc
        f += 2.0 * Lq * Lk * (64 + DV) + 2.0 * Lq * DV * DV;                               // final attention + fc on Lq rows
    } else {
        f += 2.0 * Lq * Lk * (64 + DV) + 2.0 * Lq * DV * DV;
    }
    f += Lq * (2.0 * DV * 9 * n->MID + 2.0 * n->MID * n->cfg.nclass);                       // FCNHead
    return f;
}
extern "C" double tdnet_flops_per_frame(const tdnet_t* n) { return n && n->finalized ? n->flops_frame : -1.0; }

extern "C" int tdnet_get_opts(const tdnet_t* n, tdnet_opts* out) {
    if (!n || !out) return td_fail("tdnet_get_opts: null argument");
    *out = n->opts;
    return 0;
}

extern "C" int tdnet_set_profiling(tdnet_t* n, int on) {
    if (!n) return td_fail("tdnet_set_profiling: null handle");
    n->prof = on != 0;
    n->nrec = 0;
    return 0;
}
// which: 0 conv/GEMM kernels, 1 attention kernels, 2 everything else, 3 the dominant kernel only (128x128-tile 3x3 igemm).
// mode : 0 -> summed device ms, 1 -> summed algorithmic FLOP, 2 -> launch count
static double prof_query(const tdnet* n, int which, int mode) {
    if (!n || !n->prof || n->nrec == 0) return -1.0;
    double ms = 0.0, fl = 0.0, cnt = 0.0;
    int domkind = 1;                                                   // the Winograd GEMM is the dominant kernel whenever it runs
    for (size_t i = 0; i < n->nrec; ++i) if (n->recs[i].dominant == 2) domkind = 2;
    for (size_t i = 0; i < n->nrec; ++i) {
        const ProfRec& r = n->recs[i];
        const bool take = which == 3 ? (r.family == 0 && r.dominant == domkind) : r.family == which;
        if (!take) continue;
        if (mode == 0) {
            hipEventSynchronize(r.e1);
            float t = 0.f;
            hipEventElapsedTime(&t, r.e0, r.e1);
            ms += t;
        }
        fl += r.flops; cnt += 1.0;
    }
    return mode == 0 ? ms : mode == 1 ? fl : cnt;
}
extern "C" double tdnet_last_ms(const tdnet_t* n, int which) { return prof_query(n, which, 0); }
extern "C" double tdnet_last_flops(const tdnet_t* n, int which) { return prof_query(n, which, 1); }
extern "C" double tdnet_last_launches(const tdnet_t* n, int which) { return prof_query(n, which, 2); }

// ---------------------------------------------------------------------------------------------------------------
// single-operator entry points (tests)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int tdnet_op_conv2d(const float* in, int H, int W, int Cin, const float* w_host, const float* bias_host, int Cout, int KS,
                               int stride, int dil, const float* resid, int act, const tdnet_opts* opts, int tile, float* out,
                               void* stream) {
    // tile < 0: the heuristic's tile for this shape; 0..CT_COUNT-1: forced (0: 128x128, 1: 64x128, 2: 128x64, 3..5: the same on the
    // two-stage pipeline) -- lets tests cover every variant
    if (KS != 1 && KS != 3) return td_fail("tdnet_op_conv2d: KS must be 1 or 3");
    if (tile >= CT_COUNT) return td_fail("tdnet_op_conv2d: tile must be < %d", CT_COUNT);
    const tdnet_opts o = opts_or_default(opts);
    ConvLayer L;
    std::vector<float> w(w_host, w_host + (size_t)Cout * Cin * KS * KS), b;
    if (bias_host) b.assign(bias_host, bias_host + Cout);
    const int pad = dil * (KS / 2);
    const long M = (long)out_size(H, KS, stride, dil, pad) * out_size(W, KS, stride, dil, pad);
    // tdnet_opts.overlap bit 1: an even-dilation Winograd conv runs as its two row-parity chunks (here one after the other)
    if (make_conv_layer(L, w, b, Cout, Cin, KS, stride, dil, act, false, M, o, tile < 0 ? -1 : tile, (o.overlap & 1) ? 2 : 1)) return -1;
    int rc = run_conv(nullptr, L, in, H, W, resid, out, (hipStream_t)stream);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = td_fail("tdnet_op_conv2d: device error");
    free_conv_layer(L);
    return rc;
}
// Test entry for the fp16-activation storage of tdnet_opts.precision = 1: the fp32 arguments are rounded to fp16 maps in HBM, the
// conv runs with fp16 input / residual / output (k_conv_igemm_h<.., IN16, OUT16>), and the fp16 result is widened into out.
extern "C" int tdnet_op_conv2d_f16io(const float* in, int H, int W, int Cin, const float* w_host, const float* bias_host, int Cout,
                                     int KS, int stride, int dil, const float* resid, int act, int tile, float* out, void* stream) {
    if (KS != 1 && KS != 3) return td_fail("tdnet_op_conv2d_f16io: KS must be 1 or 3");
    if (Cin % 64) return td_fail("tdnet_op_conv2d_f16io: Cin must be a multiple of 64");
    // tile 16 / 17 / 18 / 19: the LDS-DMA kernel with 128 / 192 / 256-row tiles, 256 x 256 (td_conv_hd.h); 20 / 21: 128 rows on a ring of
    // four / two LDS buffers whatever the grid (16 chooses by the grid); 22: 128 rows, eight waves; 23 / 26: the same on row images with one
    // barrier per super-step / per K step only; 24 / 25: 192 rows likewise; 27 / 28 / 29: 256 / 192 / 128 rows in the early-landing form
    // only; -1: the heuristic (DMA kernel where it applies)
    const bool no_rowimg = tile >= 48 && tile <= 61;                   // 48 + code: the same tile, tap-by-tap staging (k_conv_dma_h) instead of row images
    if (no_rowimg) tile -= 32;
    static const int code_of_tile[14] = {CD_128, CD_192, CD_256, CD_256x256, CD_128_4BUF, CD_128_2BUF, CD_128_8W,            // 16 .. 22
                                         CD_128_SUPER, CD_192_SUPER, CD_192_STEP, CD_128_STEP, CD_256_EARLY, CD_192_EARLY, CD_128_EARLY};   // 23 .. 29
    static const int code_of_tile_p[6] = {CD_128_P, CD_192_P, CD_256_P,        // 31 .. 33: row images with four dedicated loader waves (k_conv_dma_h3p)
                                          CD_128_N, CD_192_N, CD_256_N};       // 34 .. 36: narrow tiles, rows x 64 channels (k_conv_dma_h3n)
    const int force_rh = tile >= 16 && tile <= 29 ? code_of_tile[tile - 16] : tile == 30 ? CD_W64 : tile >= 31 && tile <= 36 ? code_of_tile_p[tile - 31] : 0;   // 30: the weights-resident 64 -> 64 kernel
    if (force_rh) tile = force_rh == CD_W64 ? CT_128x64 : CT_128x128_DEEP;
    if (tile >= CT_COUNT) return td_fail("tdnet_op_conv2d_f16io: tile must be < %d or 16..36 (+ 32 for 16..29)", CT_COUNT);
    hipStream_t s = (hipStream_t)stream;
    tdnet_opts o = opts_or_default(nullptr);
    o.precision = 1;
    ConvLayer L;
    std::vector<float> w(w_host, w_host + (size_t)Cout * Cin * KS * KS), b;
    if (bias_host) b.assign(bias_host, bias_host + Cout);
    const int pad = dil * (KS / 2);
    const int Ho = out_size(H, KS, stride, dil, pad), Wo = out_size(W, KS, stride, dil, pad);
    if (make_conv_layer(L, w, b, Cout, Cin, KS, stride, dil, act, false, (long)Ho * Wo, o, tile < 0 ? -1 : tile)) return -1;
    L.in16 = L.out16 = true;
    L.rowimg_off = no_rowimg;
    if (force_rh == CD_256x256 && L.CoutPad % 256) { free_conv_layer(L); return td_fail("tdnet_op_conv2d_f16io: the 256 x 256 tile needs Cout padded to a multiple of 256"); }
    if (force_rh == CD_W64 ? !conv_dma_w64_supports(Cin, Cout, L.CoutPad, KS, stride, dil, pad) : (force_rh && !conv_dma_supports(Cin, Cout, KS, L.tile))) {
        free_conv_layer(L);
        return td_fail("tdnet_op_conv2d_f16io: this shape cannot run on the LDS-DMA kernel");
    }
    if (force_rh) L.rh = force_rh;
    else if (tile < 0 && Cout >= 128 && conv_dma_supports(Cin, Cout, KS, L.tile)) L.rh = conv_dma_pick_rh((long)Ho * Wo, Cout, L.CoutPad % 256 == 0);
    _Float16 *hin = nullptr, *hres = nullptr, *hout = nullptr;
    const long nin = (long)H * W * Cin, nout = (long)Ho * Wo * Cout;
    auto cleanup = [&]() {                                             // one release path, also for the error returns
        for (_Float16* q : {hin, hout, hres}) if (q) hipFree(q);
        free_conv_layer(L);
    };
    if (dev_alloc(&hin, (size_t)nin) || dev_alloc(&hout, (size_t)nout) || (resid && dev_alloc(&hres, (size_t)nout))) { cleanup(); return -1; }
    TD_LAUNCH(k_f2h, dim3(td_grid_for(nin)), dim3(256), 0, s, in, hin, nin);
    if (resid) TD_LAUNCH(k_f2h, dim3(td_grid_for(nout)), dim3(256), 0, s, resid, hres, nout);
    int rc = run_conv(nullptr, L, (const float*)hin, H, W, (const float*)hres, (float*)hout, s);
    TD_LAUNCH(k_h2f, dim3(td_grid_for(nout)), dim3(256), 0, s, (const _Float16*)hout, out, nout);
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) rc = td_fail("tdnet_op_conv2d_f16io: device error");
    cleanup();
    return rc;
}
extern "C" int tdnet_op_stem(const float* img, int H, int W, const float* w_host, const float* bias_host, const tdnet_opts* opts,
                             float* out, void* stream) {
    const tdnet_opts o = opts_or_default(opts);
    hipStream_t s = (hipStream_t)stream;
    const int H1 = (H - 1) / 2 + 1, W1 = (W - 1) / 2 + 1;
    ConvLayer L;
    std::vector<float> w(w_host, w_host + 64 * 3 * 49), b;
    if (bias_host) b.assign(bias_host, bias_host + 64);
    if (make_conv_layer(L, w, b, 64, 3, 7, 2, 1, 1, true, (long)H1 * W1, o)) return -1;
    float *img4 = nullptr, *s1 = nullptr;
    if (dev_alloc(&img4, (size_t)H * W * 4) || dev_alloc(&s1, (size_t)H1 * W1 * 64)) return -1;
    run_stem_pre(nullptr, img, H, W, img4, s, o.fusion);
    run_conv(nullptr, L, img4, H, W, nullptr, s1, s);
    run_maxpool(nullptr, s1, H1, W1, 64, out, s, o.fusion);
    TD_HIP(hipStreamSynchronize(s));
    TD_HIP(hipGetLastError());
    hipFree(img4); hipFree(s1);
    free_conv_layer(L);
    return 0;
}
extern "C" int tdnet_op_streams_share_queue(void* stream_a, void* stream_b, int* shared) {
    if (!shared) return td_fail("tdnet_op_streams_share_queue: shared is NULL");
    *shared = 0;
    if (stream_a == stream_b) { *shared = 1; return 0; }
#ifndef TD_EMU
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) {
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        return td_fail("tdnet_op_streams_share_queue: hipEventCreate failed");
    }
    bool sh = false;
    const int rc = streams_share_a_queue((hipStream_t)stream_a, (hipStream_t)stream_b, e0, e1, e2, &sh);
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    *shared = sh ? 1 : 0;
    return rc;
#else
    return 0;
#endif
}
extern "C" int tdnet_op_attention(const float* q, const float* k, const float* vp, const float* bias, const float* resid, int Lq,
                                  int Lk, int DV, int online, const float* ln_g, const float* ln_b, float* ln_out, float* out,
                                  void* stream) {
    if (Lk < 1 || Lq < 1) return td_fail("tdnet_op_attention: empty input");
    hipStream_t s = (hipStream_t)stream;
    const bool padded = (online & 32) != 0;                            // online | 32: the caller's vp already has the padding rows (probes that time the kernel)
    const bool slices = (online & 64) != 0;                            // online | 64: DV = 512 as two 256-channel slices in one launch (the chain's cached-frame steps)
    online &= ~(32 | 64);
    // arguments are validated BEFORE anything is allocated; every later exit goes through cleanup()
    if (ln_out && (!ln_g || !ln_b)) return td_fail("tdnet_op_attention: ln_out needs ln_g and ln_b");
    if (online != 16 && (online < 0 || online > 2)) return td_fail("tdnet_op_attention: online must be 0, 1, 2 or 16");
    float *part = nullptr, *mean = nullptr, *rstd = nullptr, *vpad = nullptr;
    _Float16* vt = nullptr;                                            // online == 16: the fp16-MFMA kernel of tdnet_opts.precision = 1 (td_attn_h.h)
    auto cleanup = [&]() {
        for (float* q2 : {part, mean, rstd, vpad}) if (q2) hipFree(q2);
        if (vt) hipFree(vt);
    };
    int rc = 0;
    if (ln_out && (dev_alloc(&part, (size_t)2 * attn_strips(Lq, DV) * DV) || dev_alloc(&mean, DV) || dev_alloc(&rstd, DV))) rc = -1;   // + plane LayerNorm of the result from the epilogue's strip statistics
    if (!rc && online == 16 && dev_alloc(&vt, (size_t)DV * attn_lkpad(Lk))) rc = -1;
    if (!rc && online != 16 && !padded && attn_vp_rows(Lk) != Lk) {    // the kernels' contract: V' padded to attn_vp_rows(Lk) zero rows
        const size_t rows = (size_t)attn_vp_rows(Lk);
        if (dev_alloc(&vpad, rows * DV)) rc = -1;
        else if (hipMemsetAsync(vpad, 0, rows * DV * sizeof(float), s) != hipSuccess ||
                 hipMemcpyAsync(vpad, vp, (size_t)Lk * DV * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = td_fail("tdnet_op_attention: copy failed");
        else vp = vpad;
    }
    if (!rc) rc = run_attention(nullptr, q, k, vp, bias, resid, Lq, Lk, DV, out, s, online == 16 ? 1 : online, part, vt, slices);
    if (!rc && ln_out) run_layernorm(nullptr, out, Lq, DV, ln_g, ln_b, part, mean, rstd, ln_out, s, attn_strips(Lq, DV));
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) rc = td_fail("tdnet_op_attention: device error");
    cleanup();
    return rc;
}
extern "C" int tdnet_op_layernorm_hw(const float* x, int HW, int C, const float* g, const float* b, float* out, void* stream) {
    if (C % 4 || (C / 4 <= 256 ? 256 % (C / 4) != 0 : C / 4 > 512)) return td_fail("tdnet_op_layernorm_hw: C must be one of 4*{1,2,4,...,256} or 2048");
    float *part = nullptr, *mean = nullptr, *rstd = nullptr;
    if (dev_alloc(&part, (size_t)2 * 512 * C) || dev_alloc(&mean, C) || dev_alloc(&rstd, C)) return -1;
    run_layernorm(nullptr, x, HW, C, g, b, part, mean, rstd, out, (hipStream_t)stream);
    TD_HIP(hipStreamSynchronize((hipStream_t)stream));
    TD_HIP(hipGetLastError());
    hipFree(part); hipFree(mean); hipFree(rstd);
    return 0;
}
extern "C" int tdnet_op_ppm(const float* c4, int h, int w, const float* w_host, const float* b_host, int path_num, int pid, float* z,
                            void* stream) {
    const int C = 512, FS = C / (path_num * 4);
    if (path_num != 2) return td_fail("tdnet_op_ppm: path_num must be 2 (td4 passes path_num//2, td2 passes 2)");
    std::vector<float> pw((size_t)4 * FS * C), pb((size_t)4 * FS);
    for (int j = 0; j < 4; ++j)
        for (int o = 0; o < FS; ++o) {
            for (int c = 0; c < C; ++c) pw[((size_t)j * C + c) * FS + o] = w_host[((size_t)j * 128 + pid * FS + o) * C + c];
            pb[j * FS + o] = b_host[j * 128 + pid * FS + o];
        }
    float *dw = nullptr, *db = nullptr, *rowpart = nullptr, *pooled = nullptr, *ppmfeat = nullptr;
    if (upload(&dw, pw) || upload(&db, pb)) return -1;
    if (dev_alloc(&rowpart, (size_t)h * 36 * C) || dev_alloc(&pooled, 50 * C) || dev_alloc(&ppmfeat, 50 * FS)) return -1;
    run_ppm(nullptr, c4, h, w, C, C / 2, FS, dw, db, pid, rowpart, pooled, ppmfeat, z, (hipStream_t)stream);
    TD_HIP(hipStreamSynchronize((hipStream_t)stream));
    TD_HIP(hipGetLastError());
    for (float* q : {dw, db, rowpart, pooled, ppmfeat}) hipFree(q);
    return 0;
}
extern "C" int tdnet_op_upsample(const float* in, int C, int h, int w, int H, int W, float* out, void* stream) {
    launch_upsample(in, C, h, w, H, W, out, (hipStream_t)stream);
    TD_HIP(hipStreamSynchronize((hipStream_t)stream));
    TD_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// tuning / roofline hooks (not on the product path)
// ---------------------------------------------------------------------------------------------------------------
// Pure-MFMA loop: the practical fp32-MFMA ceiling of THIS chip at its sustained clock (4 independent accumulators per
// wave, `waves_per_simd` waves per SIMD, no memory traffic).
TD_KERNEL void k_mfma_peak(float* out, int iters) {
    f32x16 a0, a1, a2, a3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 1.f; a2[r] = 2.f; a3[r] = 3.f; }
    float x = 1.0f + (float)(threadIdx.x & 7) * 1e-3f, y = 1.0f - (float)(threadIdx.x & 3) * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = td_mfma32(x, y, a0); a1 = td_mfma32(y, x, a1); a2 = td_mfma32(x, x, a2); a3 = td_mfma32(y, y, a3);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[0] = s;                      // keep the accumulators live
}
// returns achieved TFLOP/s (fp32 MFMA) or <0
extern "C" double tdnet_bench_mfma_peak(int waves_per_simd, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* d = nullptr;
    if (hipMalloc((void**)&d, 256) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;            // 256 CUs x (4 SIMDs = one 256-thread block) x waves_per_simd
    TD_LAUNCH(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, 16);
    hipEventRecord(e0, s);
    TD_LAUNCH(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, iters);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    const double flop = (double)blocks * 4 /*waves*/ * (double)iters * 32 /*mfma per iter*/ * 4096.0;
    return ms > 0.f ? flop / (ms * 1e-3) / 1e12 : -1.0;
}
// Average device time (ms, HIP events on `stream`) of `iters` launches of one conv configuration on random data.
extern "C" double tdnet_bench_conv(int H, int W, int Cin, int Cout, int KS, int stride, int dil, int tile, int iters,
                                   const tdnet_opts* opts, void* stream) {
    const tdnet_opts o = opts_or_default(opts);
    if ((KS != 1 && KS != 3) || Cin % 32 || tile < -1 || tile >= CT_COUNT) { td_fail("tdnet_bench_conv: bad arguments"); return -1.0; }
    hipStream_t s = (hipStream_t)stream;
    ConvLayer L;
    std::vector<float> w((size_t)Cout * Cin * KS * KS), x((size_t)H * W * Cin), b(Cout, 0.1f);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : w) v = rnd() * 0.05f;
    for (auto& v : x) v = rnd();
    const int pad_ = dil * (KS / 2);
    const long M_ = (long)out_size(H, KS, stride, dil, pad_) * out_size(W, KS, stride, dil, pad_);
    if (make_conv_layer(L, w, b, Cout, Cin, KS, stride, dil, 1, false, M_, o, tile)) return -1.0;   // tile -1: the heuristic's choice for this M
    float *din = nullptr, *dout = nullptr;
    if (upload(&din, x)) return -1.0;
    const int Ho = out_size(H, KS, stride, dil, L.pad), Wo = out_size(W, KS, stride, dil, L.pad);
    if (dev_alloc(&dout, (size_t)Ho * Wo * Cout)) return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    tdnet tmp;                                                        // only carries the Winograd workspace for run_conv
    if (L.wino) {
        const size_t T = (size_t)wino_tiles(H, W, dil, L.wino), nb = (size_t)(L.wino + 2) * (L.wino + 2);
        tmp.wino_v_floats = nb * (T + L.wino_pad) * Cin; tmp.wino_m_floats = nb * (T + L.wino_pad) * Cout;
        if (dev_alloc(&tmp.wino_v, tmp.wino_v_floats) || dev_alloc(&tmp.wino_m, tmp.wino_m_floats)) return -1.0;
    }
    tdnet* ws = L.wino ? &tmp : nullptr;
    for (int i = 0; i < 2; ++i) run_conv(ws, L, din, H, W, nullptr, dout, s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) run_conv(ws, L, din, H, W, nullptr, dout, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(din); hipFree(dout);
    if (tmp.wino_v) hipFree(tmp.wino_v);
    if (tmp.wino_m) hipFree(tmp.wino_m);
    free_conv_layer(L);
    return ms / iters;
}
