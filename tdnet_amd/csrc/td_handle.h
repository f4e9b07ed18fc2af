// td_handle.h -- host-side types of libtdnet_hip.so: error reporting, the architecture description, a device conv layer, the WEIGHT BLOCK
// (folded + packed weights of all paths, shared by every handle created from it) and the HANDLE (workspace, K/Q/V FIFO, streams).
//
// Split of the former td_model.hip monolith (round 5):
//   td_handle.h    this file
//   td_weights.h   strict state_dict inventory, BN folding (fp64), packing, upload, the row-parity plan, workspace allocation
//   td_launch.h    one launch helper per operator (conv / Winograd conv / attention / LayerNorm / pyramid / stem / classifier / upsample)
//   td_frame.h     the per-frame kernel sequence: FIFO, cache-only attention chain, row-parity chains, encode / finish, stream placement
//   td_ops_test.h  single-operator entry points for the tests + roofline / tuning probes (not on the product path)
//   td_model.hip   the translation unit: the C ABI of include/tdnet.h
#pragma once
#include "../../include/tdnet.h"
#include "td_device.h"
#include "td_conv.h"
#include "td_conv_h.h"
#include "td_conv_hd.h"
#include "td_conv_ad.h"
#include "td_wino.h"
#include "td_gemm.h"
#include "td_gemm_dma.h"
#include "td_gemm_b3.h"
#include "td_conv_ad_b3.h"
#include "td_attn.h"
#include "td_attn_h.h"
#include "td_attn_b3.h"
#include "td_misc.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <atomic>
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
static int out_size(int n, int KS, int stride, int dil, int pad) { return (n + 2 * pad - dil * (KS - 1) - 1) / stride + 1; }

// ---------------------------------------------------------------------------------------------------------------
// device conv layer
// ---------------------------------------------------------------------------------------------------------------
struct ConvLayer {
    int Cin = 0, Cout = 0, KS = 1, stride = 1, dil = 1, pad = 0, act = 0;
    bool stem = false;
    bool stem_rows = false;                                            // the 7x7 fp32 stem on the packed-row image (td_conv_ad.h STEM = 2; tdnet_opts.fusion bit 65536)
    bool h16 = false;                                                  // fp16-MFMA operands (td_conv_h.h)
    int wino_pad = 0;                                                  // padding rows per Winograd plane (fusion bit 64)
    bool adirect = false;                                              // Cout <= 64: A operand straight from global (td_conv_ad.h, fusion bit 32)
    bool in16 = false, out16 = false;                                  // h16 only: the input (+ residual) / output map is stored as fp16 in HBM
    int rh = 0;                                                        // h16 + in16: != 0 -> the LDS-DMA kernel with 64 rh rows per tile (td_conv_hd.h); M_out: its output pixels
    long M_out = 0;
    bool rowimg_off = false;                                           // tdnet_opts.fusion bit 2048: keep the tap-by-tap LDS-DMA kernel
    int pers = 1;                                                      // tdnet_opts.gemm_persistent of the owning handle
    int chunks = 1;                                                    // > 1: run as that many row-parity chunks (tdnet_opts.overlap bit 1); the GEMM tile is picked for T / chunks rows
    int b3 = 0;                                                        // != 0: d_wp holds the three bf16 parts of the weights (td_gemm_b3.h gemm_b3_pack; tdnet_opts.precision = 2) and the GEMM runs on k_gemm_b3
    bool gdma = false;                                                 // the Winograd GEMMs on the LDS-DMA-fed kernel (td_gemm_dma.h; tdnet_opts.overlap bit 8)
    int vw = 0;                                                        // != 0: the low-register F(4x4) transform kernels with vw channels per lane (td_wino.h k_wino4_*_c)
    int wino = 0;                                                      // Winograd output tile edge m (0 = direct, 4 = F(4x4,3x3)): d_wp = 36 packed 1x1 weight sets (td_wino.h)
    float* d_zero = nullptr;                                           // zero bias for the batched GEMM pass
    ConvTile tile = CT_128x128;
    int CoutPad = 0, nsteps = 0;
    float* d_wp = nullptr;
    float* d_bias = nullptr;
    double flops_per_pixel() const { return 2.0 * Cout * (stem ? 3.0 * KS * KS : (double)Cin * KS * KS); }
};

// Per-handle kernel configuration (include/tdnet.h tdnet_opts); nothing here is process-wide: two handles in one process may differ.
static tdnet_opts opts_or_default(const tdnet_opts* o) {
    tdnet_opts d;
    tdnet_opts_default(&d);
    if (!o) return d;
    d = *o;
    d.winograd = d.winograd <= 0 ? 0 : (d.winograd == 2 || d.winograd >= 4) ? 4 : 3;   // 1 / 2 were F(2x2,3x3) (removed in round 5): the F(4x4) forms of the same scope
    d.precision = d.precision < 0 ? 0 : d.precision > 3 ? 1 : d.precision;   // 0 fp32 MFMA, 1 fp16 MFMA, 2 fp32-accurate GEMMs on the bf16 MFMA (td_gemm_b3.h), 3 = 2 at any GEMM size (tests)
    d.pipeline = d.pipeline ? 1 : 0;
    d.gemm_persistent = d.gemm_persistent < 0 ? 0 : d.gemm_persistent;
    d.attention = d.attention < 0 ? 0 : d.attention > 2 ? 2 : d.attention;
    d.overlap = d.overlap < 0 ? 0 : d.overlap & TDNET_OVERLAP_MASK;
    if (((d.overlap >> 4) & 3) == 3) d.overlap &= ~0x30;
    d.reserved0 = 0;
    for (int& r : d.reserved) r = 0;
    return d;
}

// ---------------------------------------------------------------------------------------------------------------
// weight block + handle
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

// Everything a model owns that does NOT change from frame to frame: the host state_dict until it is finalized, then the BN-folded,
// packed, uploaded layers of every path.  One block serves any number of handles (tdnet_create_shared: the samples of a batch, the
// two lanes of a frame-pipelined clip, clips sharing a GPU) -- the reference's batch shares ONE nn.Module's parameters the same way
// (td4_psp18.py:216-229).  Reference-counted; freed with the last handle, whatever the destroy order.
struct TdWeights {
    std::atomic<int> refs{1};                                          // handles on this block; tdnet_create_shared / tdnet_destroy may run on different host threads
    int device = 0;
    std::vector<BlockSpec> bspec;
    std::map<std::string, std::vector<float>> sd;                      // host state_dict until finalize
    std::map<std::string, size_t> expected;                            // name -> element count
    bool finalized = false;
    std::vector<PathLayers> paths;
    // Row-parity chains (tdnet_opts.overlap bit 1): the trailing run of even-dilation convs of the backbone starts at conv seg_conv
    // (0: conv1, 1: conv2) of block seg_block (-1: off) -- td_weights.h plan_chains
    int seg_block = -1, seg_conv = 0;
    bool act16 = false;                                                // precision = 1: the maps between the backbone's convs are fp16 in HBM
    double flops_frame = 0.0;
    size_t device_bytes = 0;                                           // HBM held by the block (packed weights, biases, affine maps)
};

struct tdnet {
    TdWeights* const wt;                                               // never null; shared between handles (ref-counted)
    tdnet_cfg cfg;
    tdnet_opts opts;                                                   // per-handle kernel configuration (never process-wide); handles of one block share it
    int P = 0, DV = 0, MID = 0, FIFO = 0, C = 512, SC = 64;            // C = backbone output channels, SC = stem output channels
    bool deep = false;
    int H = 0, W = 0, H1 = 0, W1 = 0, H2 = 0, W2 = 0, h = 0, w = 0, hk = 0, wk = 0, Lq = 0, Lk = 0;
    // views of the weight block under the names the frame code uses
    std::vector<BlockSpec>& bspec;
    std::map<std::string, std::vector<float>>& sd;
    std::map<std::string, size_t>& expected;
    bool& finalized;
    std::vector<PathLayers>& paths;
    int& seg_block; int& seg_conv;
    bool& act16;
    double& flops_frame;
    bool ws_ready = false;                                             // workspace, FIFO slots, streams and events of THIS handle exist
    size_t ws_bytes = 0;                                               // HBM held by this handle alone (workspace + FIFO)
    // workspace
    float *img4 = nullptr, *s1 = nullptr, *s1b = nullptr, *bx = nullptr, *bt = nullptr, *br = nullptr, *bu = nullptr;
    float *rowpart = nullptr, *pooled = nullptr, *ppmfeat = nullptr, *z = nullptr;
    float *v_cur = nullptr, *q1 = nullptr, *q_cur = nullptr, *k1 = nullptr;
    float *vp = nullptr, *chain_a = nullptr, *chain_b = nullptr, *feat = nullptr;
    float *ln_part = nullptr, *ln_mean = nullptr, *ln_rstd = nullptr, *ln = nullptr;
    float *headmid = nullptr, *lowres = nullptr, *stage_tmp = nullptr, *logits_tmp = nullptr;
    float *wino_v = nullptr, *wino_m = nullptr;                        // Winograd workspaces [36][T][Cin] / [36][T][Cout]
    size_t wino_v_floats = 0, wino_m_floats = 0;
    size_t stage_tmp_floats = 0;
    std::vector<CacheSlot> slots;
    std::vector<int> fifo;                                             // slot ids, oldest first
    int last_slot = -1;
    int pending_slot = -1;                                             // cache entry of an encoded, not yet propagated frame
    int pending_pos = -1;
    // cache-only work (V' GEMMs + the two cached-frame attention steps) runs on a side stream under the backbone
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;                 // Encoding's q / k projections beside w_vs (fusion bit 1)
    // Row-parity chains: chain 0 runs on the forward's stream with wino_v / wino_m, chain 1 on `chain2` with wino_v2 / wino_m2.  The
    // chains may drift apart by more than a block and the channel count changes inside the run, so no map of the run is written in
    // place or shared between blocks: block b owns seg_t[b] (conv1 output), seg_r[b] (downsample output) and seg_x[b] (block output).
    hipStream_t chain2 = nullptr;
    hipEvent_t ev_cfork = nullptr, ev_cjoin = nullptr;
    float *wino_v2 = nullptr, *wino_m2 = nullptr;
    std::vector<float*> seg_t, seg_r, seg_x;
    std::vector<hipStream_t> probe_streams;                            // TDNET_PROBE_EXTRA_STREAMS (-DTDNET_TIMING_PROBES builds only)
    std::vector<hipStream_t> retired_streams;                          // chain2 candidates that shared the caller's hardware queue (place_chain_stream)
    std::vector<void*> placed_for;                                     // the caller streams chain2 has been checked against (place_chain_stream): once per stream
    int chain_replaced = 0;
    float* c4 = nullptr;                                              // backbone output of the last frame (bx, or br in the fp16-activation mode)
    _Float16* vt16 = nullptr;                                         // fp16 attention: V' re-tiled [LkPad / 8][DV][8]; precision 2: its three bf16 parts (3x that, td_attn_b3.h)
    bool ln_pending = false;                                          // the `ln` map of the last frame was not materialised (fusion bit 4)
    int ln_path = 0;
    bool feat_is_vcur = false;                                        // warm-up frame (td4_psp18.py:142-143): the "feat" stage IS v_cur (no copy is made)
    bool failed = false;                                              // a launch helper reported an error during the current forward
    bool prof = false;
    std::vector<ProfRec> recs;
    size_t nrec = 0;
    int launches = 0;                                                  // kernel launches + device copies enqueued by the current frame (td_launch.h TD_COUNTED)

    explicit tdnet(TdWeights* w)
        : wt(w), bspec(w->bspec), sd(w->sd), expected(w->expected), finalized(w->finalized), paths(w->paths), seg_block(w->seg_block),
          seg_conv(w->seg_conv), act16(w->act16), flops_frame(w->flops_frame) {}
    tdnet(const tdnet&) = delete;
    tdnet& operator=(const tdnet&) = delete;
};

// device allocations of a handle / of a weight block are counted (bytes), so that the cost of an extra lane is a number (tdnet_memory_bytes)
static thread_local size_t* g_alloc_counter = nullptr;
template <typename T>
static int dev_alloc(T** p, size_t count) {
    TD_HIP(hipMalloc((void**)p, count * sizeof(T)));
    if (g_alloc_counter) *g_alloc_counter += count * sizeof(T);
    return 0;
}
static int upload(float** d, const std::vector<float>& v) {
    TD_TRY(dev_alloc(d, v.size()));
    TD_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
struct AllocScope {                                                   // RAII: allocations inside the scope are added to *counter
    size_t* prev;
    explicit AllocScope(size_t* counter) : prev(g_alloc_counter) { g_alloc_counter = counter; }
    ~AllocScope() { g_alloc_counter = prev; }
};
