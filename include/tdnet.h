/*
 * tdnet.h -- C ABI of the MI355X-native TDNet per-frame inference hot path (libtdnet_hip.so).
 *
 * The reference's boundary for this path is a Python nn.Module API, not a C API:
 *     model = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=...)      Testing/test.py:26
 *     out   = model(image, pos_id=i % path_num)                               Testing/test.py:53
 * tdnet_amd/model/{td4_psp18,td2_psp50,pspnet}.py keep that Python API and call the entry points below through
 * ctypes; INTEGRATION.md shows the stub.  Plain pointers and sizes only -- no torch types cross this line.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; tdnet_last_error() gives the message (thread-local).
 *   - "dev" pointers are device (HBM) pointers owned by the caller; the handle owns weights, workspace and
 *     the K/Q/V FIFO.  One handle per video stream and per GPU; a handle is not thread-safe.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  The frame calls (tdnet_forward*, tdnet_encode,
 *     tdnet_propagate*, tdnet_cache_*) only ENQUEUE work and never synchronise with the host -- with one exception a caller can
 *     remove: a handle whose frame uses a second internal stream (row-parity chains) checks ONCE per caller stream that this stream
 *     sits on another hardware queue than the caller's (two 40-us spin kernels, a host synchronisation of both streams).
 *     tdnet_warmup(h, stream) does that check explicitly; a frame call on a stream tdnet_warmup has not seen does it lazily
 *     (skipped while the stream is being captured into a hipGraph).  Every caller stream is checked ONCE per handle (the handle remembers
 *     the streams it has seen; alternating between two streams does not repeat the check), for at most 8 distinct streams.  tdnet_finalize_weights, tdnet_create_shared, tdnet_get_stage
 *     synchronise (as do the test library's tdnet_op_* / tdnet_bench_* entries, include/tdnet_test.h).
 *   - all tensors are fp32.  Image in / logits out are NCHW like the reference; internal layout is NHWC.
 */
#ifndef TDNET_H
#define TDNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tdnet tdnet_t;

typedef struct tdnet_cfg {
    int32_t model;      /* 4 = td4 (Testing/model/pspnet/td4_psp18.py:29-120), 2 = td2 (td2_psp50.py:29-96),
                           1 = pspnet, the stateless comparison model (pspnet.py:31-115)                             */
    int32_t backbone;   /* 18 / 34 (BasicBlock, resnet.py:218-236), 50 / 101 (Bottleneck + deep stem, :62-111,122-131)     */
    int32_t nclass;     /* 19 for Cityscapes (test.py:26)                                                            */
    int32_t height;     /* input H; the LayerNorm affine is [ceil(H/8), ceil(W/8)] (td4_psp18.py:107-110)            */
    int32_t width;      /* input W                                                                                   */
    int32_t device;     /* HIP device ordinal                                                                        */
} tdnet_cfg;

/* Per-handle kernel configuration.  Nothing in this library is process-wide: two handles in one process may differ.
 * tdnet_opts_default() fills the defaults; fields left 0 by a caller that memset()s the struct select the plain variants. */
#define TDNET_WINOGRAD_DEFAULT 3
#define TDNET_ATTENTION_DEFAULT 2
#define TDNET_FUSION_DEFAULT 2072614  /* 2 | 4 | 32 | 8192 | 32768 | 65536 | 131072 | 262144 | 524288 | 1048576; bit 1048576 (fp32 / precision 2) since round 6: 273.9 -> 275.4 frames/s at C3, bit-identical (profiles/r06al_*); bit 524288 (precision 2 only) since round 6: 322.9 -> 339.6 frames/s at C3 with precision 2, 441.8 -> 458.9 at 769x1537, td2-psp34 720x960 446.9 -> 469.0 (profiles/r06z_*); bit 262144 since round 6: 276.4 -> 277.2 frames/s at C3, 393.6 -> 394.3 at 769x1537, one launch fewer, bit-identical (profiles/r06p_*); bit 32 since the end of round 3: 270.1 -> 274.4 frames/s at C3, twice on one box (profiles/r03y_*);
                                       bits 8192 and 32768 (precision 1 only) since round 4: 1042 -> 1053 -> (see DESIGN) frames/s at 720x960 fp16, bit-identical
                                       (profiles/r04j_*, r04x_*); bit 65536 (fp32 only) since round 5: the 7x7 stem 168 -> 125 us at 1024x2048, frames/s +0.2 % (td4
                                       1024x2048, where the stem runs beside the cache-only attention chain) ... +1.5 % (td2 1024x2048) (profiles/r05j_*);
                                       bit 131072 (precision 1 only) since round 5: 54 -> 50 launches and 1131 -> 1162 frames/s at 720x960 fp16, 722 -> 729 at 1024x2048 fp16,
                                       bit-identical (profiles/r05l_*) */
#define TDNET_OVERLAP_DEFAULT 41   /* row-parity chains with 4 channels per lane (+2 %) on the LDS-DMA-fed GEMM (+0.9 %): profiles/r03a_*, r03m_* */
typedef struct tdnet_opts {
    int32_t winograd;        /* conv algorithm: 0 = direct implicit GEMM everywhere, 3 (default) = Winograd F(4x4,3x3) for the stride-1 3x3
                                convs with Cin, Cout >= 128 (ResNet layers 2-4 + FCN head), 4 = F(4x4,3x3) for every stride-1 3x3 (test
                                hook).  All fp32.  (1, 2 were F(2x2,3x3), removed in round 5; they are read as 3, 4.)                  */
    int32_t precision;       /* 0 = exact fp32 MFMA (default; the headline), 1 = fp16 MFMA with fp32 accumulation (BASELINE config 5),
                                2 = OPT-IN, fp32-ACCURATE on the 16x faster bf16 MFMA: the large GEMMs of the frame (the 36 batched GEMMs of every
                                    Winograd F(4x4) conv, the stride-1 1x1 convs of >= 256 tiles of 256 x 128) take their fp32 operands as the exact
                                    sum of three bf16 values and multiply six bf16 products with fp32 accumulation (td_gemm_b3.h): every product
                                    to 2^-26 relative, results not bit-identical to the fp32 MFMA's but held to the SAME gates (1e-3 + tie band on
                                    the calibrated clips, 4x / 3x of the CPU's own error on un-calibrated init: tests/test_gpu_b3.py).  Everything
                                    else (attention, direct convs, transforms, storage) is the fp32 path unchanged.  td4-psp18 1024x2048: 276 ->
                                    303 frames/s, the dominant GEMM 306 -> 187 us (profiles/r06*),
                                3 = TEST HOOK: 2 with the split kernel at ANY GEMM size (small maps in the tests)                        */
    int32_t pipeline;        /* conv software pipeline: 0 = one-stage prefetch, 1 (default) = two-stage                           */
    int32_t gemm_persistent; /* 1 (default) = stride-1 1x1 convs and the Winograd GEMMs on the persistent multi-tile GEMM kernel,
                                0 = one tile per workgroup on the conv kernel, n > 1 = persistent with the grid forced to n (tests)  */
    int32_t reserved0;       /* must be 0 (rounds 1-4: `stagger`, a start delay of co-resident workgroups; measured neutral, removed) */
    int32_t attention;       /* 0 = exact two-pass softmax (row maxima first), 1 = single pass, lazily moved reference, 2 (default) = the same pipelined to one barrier per key tile        */
    int32_t fusion;          /* bit mask of launch-level fusions / overlaps, each measured on its own; default 2|4|32 (+ 8192|32768 in fp16 mode):
                                1 = Encoding's q / k projections (w_qs, w_ks: small, latency-bound) on the side stream beside w_vs,
                                2 = LayerNorm strip statistics written by the attention epilogue (no separate pass over the map),
                                4 = LayerNorm normalisation applied inside the head's Winograd input transform (no `ln` map in HBM),
                                8 = (retired, ignored: the pyramid row sums now read the map once, td_misc.h k_ppm_rowsum),
                                16 = stem: 4-pixel vectorised layout change and 2-output max-pool,
                                32 = Cout <= 64 convs (layer1, the stems) read their A operand straight from global memory in MFMA
                                     fragment layout instead of staging it through LDS (td_conv_ad.h),
                                64 = the 36 planes of the Winograd workspaces V / M padded by 24 rows each (an unpadded plane is a
                                     power of two bytes: 36 concurrent streams on the same HBM channels),
                                128 = precision 1 only: keep the convs that read fp16 maps on the register-staged kernel (td_conv_h.h)
                                     instead of the LDS-DMA kernel (td_conv_hd.h) -- A/B of the round-3 kernel,
                                256 = the 4-pixel vectorised layout change of bit 16 alone (without its 2-output max-pool),
                                512 = the cached-frame attention steps of td4's propagation chain as ONE 512-channel launch (default: two
                                     256-channel slices per launch, twice the workgroups),
                                1024 = precision 1 only: no 256 x 256 tiles in the LDS-DMA conv kernel (A/B),
                                2048 = precision 1 only: the LDS-DMA conv stages its activation operand tap by tap (k_conv_dma_h) instead of
                                     one LDS image per kernel row shared by the row's three taps (k_conv_dma_h3) -- A/B,
                                4096 = (removed in round 5, ignored: layer1 on persistent workgroups with the weights resident in LDS; no faster),
                                8192 = precision 1 only (default): the 128 / 192 x 128 tiles of the LDS-DMA conv with four dedicated LOADER waves per
                                     workgroup (k_conv_dma_h3p: the matrix waves never issue vector memory inside the K loop); bit-identical,
                                32768 = precision 1 only (default): on maps of <= 16384 output pixels the 3x3 "same" convs with <= 256 output channels run on
                                     NARROW tiles (128 / 192 rows x 64 channels, k_conv_dma_h3n: half the weight bytes per K step and CU); bit-identical.
                                     (16384 was an experiment removed in round 4 and is ignored.)  Since round 5 bit 32768 also routes ResNet layer1
                                     (64 -> 64 channels) to the narrow kernel: 11.9 -> 9.8 us isolated, 1090 -> 1121 frames/s at 720x960.
                                     (Round 5 also tried the narrow tiles on the 512-channel convs and 64- / 96-row narrow tiles: 1 - 7 % slower
                                     in the frame, removed; profiles/r05c_*.)
                                65536 = fp32, with bit 32 (round 5): the 7x7 stem reads a PACKED-ROW image -- [H + 7][~W + 9][3] floats with a zero
                                     border, written by the layout kernel -- so that the 21 (kx, channel) products of a kernel row are contiguous:
                                     a K step is one kernel row, K = 168 instead of 224 for the same 147 products (td_conv_ad.h STEM = 2).
                                131072 = precision 1 only (default, round 5): a BasicBlock's conv1 and 1x1 downsample in ONE launch where both run on the
                                     register-staged kernel (ResNet layer2.0 at 720x960), and the Encoding's five 1x1 convs in TWO launches -- value / query / key first layers
                                     side by side on z, then the query / key second layers (k_conv_igemm_h_group: blocks of up to three convs in one
                                     grid); the value conv is packed for the 64-channel tile of the others.  Same products, same order: bit-identical.
                                262144 = fp32 / precision 2 (default, round 6): the FCN head's 1x1 classifier inside its 3x3 conv's Winograd output transform
                                     (k_wino4_out_cls; td4_psp18.py:295-299): one launch fewer, the 128- / 64-channel hidden map is never written; same
                                     arithmetic in the same order: bit-identical low-resolution logits.
                                524288 = precision 2 only (default, round 6): the Cout <= 64 convs that read their A operand straight from global memory (bit 32:
                                     ResNet layer1, the packed-row 7x7 stem of bit 65536) on the bf16 MFMA with both operands as three bf16 parts
                                     (td_conv_ad_b3.h): layer1's conv 85 -> 59 us at 1024x2048, errors against fp64 at or below the fp32 kernel's; also the
                                     direct convs of 65 .. 128 output channels (the strided convs of layer2.0, a deep stem's 64 -> 128 conv) as two 64-column tiles.
                                1048576 = fp32 / precision 2 (default, round 6): the cache-only attention chain of a frame (side stream) forks in front of the backbone's first
                                     dilated block (layer3) instead of at the frame's start: it then overlaps the large Winograd GEMMs, not the stem and layer1.
                                     Same launches, same results bit for bit; +0.2 ... 1.0 % by workload; ignored in the fp16 mode (-2.6 % there). */
    int32_t overlap;         /* bit mask (default TDNET_OVERLAP_DEFAULT), on BasicBlock backbones:
                                1 = the trailing run of even-dilation convs (ResNet layers 3-4: resnet.py:181-198) is split into its
                                    even-row and odd-row halves -- a dilated conv maps a row parity onto itself, so the halves are independent
                                    chains -- on two HIP streams.  fp32 (winograd >= 3): the HBM-bound transforms of one chain run under the
                                    MFMA-bound GEMMs of the other.  (precision 1: ignored.  Round 5 ran the fp16 mode's direct convs as row
                                    classes as well -- bit-identical, 1121 -> 1050 frames/s at 720x960: removed, profiles/r05a_*.)
                                    By default only on maps of >= 24000 feature pixels (h w): td4-psp18 measured -2 % with the chains at 8192 ..
                                    18721 pixels (512x1024 .. 769x1537), +0.8 % at 25088, +1.8 % at 32768 (profiles/r05g_*),
                                2 = the low-register transform kernels for every F(4x4) conv, chained or not,
                                4 = the chains of bit 1 at ANY map size (tests, A/B),
                                8 = the Winograd GEMMs on the LDS-DMA-fed kernel (td_gemm_dma.h: no staging registers, 82 VGPRs),
                                bits 4-5 = channels per lane of the chunked transform kernels: 0 -> 1, 1 -> 2, 2 -> 4,
                                64 = PROBE HOOK, tdnet_op_conv2d / tdnet_bench_conv only (the frame ignores it): with bit 1, a conv whose dilation is
                                     a multiple of 4 as FOUR row classes mod 4 instead of two.  Round 5's Infinity-Cache residency experiment: a
                                     class's V + M is 76 MB at 1024x2048 instead of 151 MB; transforms -5 %, GEMMs +10 %, frame -1.2 % -- not adopted
                                     (tools/wino_l3_probe.py, profiles/r05a_l3_*).
                                (rounds 3-4 used bit 4 for a staggered start of the second chain and 64 for transforms riding inside the other chain's GEMM launches,
                                 128 = the next frame's cache-only chain launched at the end of this one: measured neutral to negative in
                                 rounds 3-4, removed in round 5 and ignored; DESIGN_experiments.md 4.1d, 8.)                                  */
    int32_t reserved[8];     /* must be 0 (round 4: cu_reserve / cu_mode, the CU-mask-partitioned pipeline, -2.5x, removed)        */
} tdnet_opts;
#define TDNET_OVERLAP_MASK 0x7f    /* the bits of tdnet_opts.overlap that exist: 1 | 2 | 4 | 8 | 16 | 32 | 64 */
void tdnet_opts_default(tdnet_opts* o);

/* ---- lifecycle: replaces the nn.Module constructor + load_state_dict (td4_psp18.py:32-120, :232-240) ---------- */
int  tdnet_create(const tdnet_cfg* cfg, tdnet_t** out);                                  /* default options */
int  tdnet_create_opts(const tdnet_cfg* cfg, const tdnet_opts* opts /* NULL = defaults */, tdnet_t** out);
int  tdnet_get_opts(const tdnet_t* h, tdnet_opts* out);
void tdnet_destroy(tdnet_t* h);

/* One call per state_dict entry, reference key names ("pretrained1.layer4.1.conv2.weight", ...), host fp32
 * data in the reference's own layout (conv OIHW).  Unknown names and wrong sizes are errors (strict=True,
 * td4_psp18.py:237); unused reference tensors (pretrainedN.fc.*, *.num_batches_tracked) are accepted and ignored. */
int  tdnet_set_weight(tdnet_t* h, const char* name, const float* host, size_t count);
/* Folds BN (fp64), repacks to the kernels' layouts, uploads; then allocates the handle's workspace, FIFO and streams.
 * Fails listing the first missing tensor.  Synchronises the device.                                               */
int  tdnet_finalize_weights(tdnet_t* h);
/* A further handle on the SAME weight block as `weights_of` (which must be finalized): own workspace, own K/Q/V FIFO, own
 * streams -- another video stream on this GPU, the samples 1..N-1 of a batch (the reference's batch shares one nn.Module's
 * parameters: td4_psp18.py:216-229), or the second lane of a frame-pipelined clip -- without a second copy of the packed weights
 * and without folding / packing / uploading them again.  `opts` must be NULL (inherit) or equal to the block's options: the
 * packing depends on them.  The block is reference-counted (atomically): handles may be destroyed in any order, the weights go with the
 * last.  Threading: a HANDLE is single-threaded (one host thread at a time), but tdnet_create_shared / tdnet_destroy of DIFFERENT handles
 * on one block may run concurrently on different host threads (a garbage collector's finaliser thread, say).                    */
int  tdnet_create_shared(const tdnet_t* weights_of, const tdnet_opts* opts /* NULL = inherit */, tdnet_t** out);
/* One-time placement of the handle's internal streams for frames that will arrive on `stream` (see "Conventions"): host-
 * synchronising, idempotent per stream.  Call it before capturing frames into a hipGraph or before a latency-critical first frame. */
int  tdnet_warmup(tdnet_t* h, void* stream);
/* HBM held by the weight block (*weights_bytes, shared) and by this handle alone (*handle_bytes: workspace + FIFO).  Returns the
 * number of handles currently sharing the block, <0 on error.                                                             */
int  tdnet_memory_bytes(const tdnet_t* h, size_t* weights_bytes, size_t* handle_bytes);

/* ---- the hot path: replaces model(image, pos_id) (td4_psp18.py:216-229 / td2_psp50.py:146-155) ---------------- */
/* img_nchw_dev [1,3,H,W] -> logits_nchw_dev [1,nclass,H,W].  Mutates the K/Q/V FIFO exactly like
 * buffer_contral (td4_psp18.py:123-134): frames must arrive in order with pos_id = t mod path_num.              */
int  tdnet_forward(tdnet_t* h, const float* img_nchw_dev, int pos_id, float* logits_nchw_dev, void* stream);
/* argmax over classes with first-max tie-break = output.max(1)[1] (test.py:61); labels int32 [H,W].             */
int  tdnet_argmax(tdnet_t* h, const float* logits_nchw_dev, int32_t* labels_dev, void* stream);
/* forward + argmax without materialising the full-resolution logits (labels identical to the two calls above).  */
int  tdnet_forward_labels(tdnet_t* h, const float* img_nchw_dev, int pos_id, int32_t* labels_dev, void* stream);
/* Empties the FIFO (the reference never resets between clips; needed to feed a second clip).                     */
int  tdnet_reset(tdnet_t* h);
int  tdnet_fifo_len(const tdnet_t* h);

/* ---- split frame + cache transport: path-parallel single stream (SURVEY 8e "alternative" / 8f-N4) ----------------
 * tdnet_forward == tdnet_encode followed by tdnet_propagate.  The split exists so that W GPUs can serve ONE video
 * stream, GPU g taking the frames t = g (mod W): the cache entry (q,k,v) a frame contributes to its successors
 * (Encoding pre=True, transformer.py:34-50; pushed by buffer_contral td4_psp18.py:123-134) exists after tdnet_encode,
 * is read out with tdnet_cache_export, travels to the peers (one all-gather per round over xGMI), and is inserted into
 * their FIFOs with tdnet_cache_push in frame order; tdnet_propagate then runs attention propagation + head against
 * the FIFO as it stands and commits the frame's own entry.  All calls are stream-ordered on `stream`.                */
/* backbone + pyramid slice + Encoding of one frame; leaves its cache entry pending (td4_psp18.py:138-140,153).     */
int  tdnet_encode(tdnet_t* h, const float* img_nchw_dev, int pos_id, void* stream);
/* attention propagation + LayerNorm + head + x8 upsample of the pending frame (td4_psp18.py:142-152), then FIFO push */
int  tdnet_propagate(tdnet_t* h, float* logits_nchw_dev, void* stream);
int  tdnet_propagate_labels(tdnet_t* h, int32_t* labels_dev, void* stream);
/* cache entry geometry: q,k are [Lk,dk], v is [Lk,dv] fp32                                                         */
int  tdnet_cache_dims(const tdnet_t* h, int* Lk, int* dk, int* dv);
/* copy the pending frame's entry into caller-owned device buffers                                                  */
int  tdnet_cache_export(tdnet_t* h, float* q_dev, float* k_dev, float* v_dev, void* stream);
/* append an entry computed by a peer to the FIFO (oldest entry drops out when the FIFO is full)                    */
int  tdnet_cache_push(tdnet_t* h, const float* q_dev, const float* k_dev, const float* v_dev, void* stream);

/* ---- introspection for the parity tests ------------------------------------------------------------------------ */
/* Copies an internal stage buffer of the LAST frame to host, converted to the reference's layout
 * (NCHW for maps, [L,C] for q/k/v).  names: c4 z q_cur v_cur feat ln lowres cache_q cache_k cache_v.
 * Returns the element count, or <0.                                                                              */
long tdnet_get_stage(tdnet_t* h, const char* name, float* host, size_t capacity);
/* Algorithmic FLOP (2*MAC, conv + attention matmuls) of one steady-state frame for this configuration.          */
double tdnet_flops_per_frame(const tdnet_t* h);
/* Device time of the dominant kernel family inside the last tdnet_forward (HIP events on the forward's stream):
 * which: 0 = all conv/GEMM kernels, 1 = attention kernels, 2 = everything else, 3 = the dominant kernel only
 * (fp32: the Winograd GEMMs, or the 128x128-tile 3x3 implicit-GEMM conv when Winograd is off; fp16 mode: the LDS-DMA / 128x128 3x3
 * convs), 4 = EVERY 3x3 conv that reads an fp16 map (fp16 mode: a fixed set of layers, whatever kernel each is routed to).
 * ms, <0 if profiling is off.                                                                                       */
int    tdnet_set_profiling(tdnet_t* h, int on);
double tdnet_last_ms(const tdnet_t* h, int which);
/* same selection: summed algorithmic FLOP / number of launches of that family in the last forward.              */
double tdnet_last_flops(const tdnet_t* h, int which);
double tdnet_last_launches(const tdnet_t* h, int which);
/* Kernel launches the last tdnet_forward* / tdnet_encode / tdnet_propagate* of this handle enqueued (all streams, the final x8 upsample /
 * upsample + argmax kernel included; device copies included: none in a steady-state frame).  Counted in the launch macro itself,
 * profiling on or off.                                                                                              */
int    tdnet_last_launch_count(const tdnet_t* h);

/* Do two HIP streams run on ONE hardware queue (HIP deals streams onto a small pool of queues, GPU_MAX_HW_QUEUES per priority class, and
 * reuses them; kernels of two streams on one queue run one after the other)?  Two 40-us spin kernels started together: *shared = 1 when
 * they serialise.  Synchronises both streams with the host.  The module uses it to place the streams of the samples of a batch
 * (tdnet_amd/model/_base.py); a handle applies the same test to its internal streams at its first frame.                              */
int tdnet_streams_share_queue(void* stream_a, void* stream_b, int* shared);
const char* tdnet_last_error(void);
const char* tdnet_version(void);

#ifdef __cplusplus
}
#endif
#endif
