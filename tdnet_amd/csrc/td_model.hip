// td_model.hip -- the translation unit of libtdnet_hip.so: the C ABI of include/tdnet.h, and nothing else.  The host code behind it lives in the
// headers this file pulls in (td_frame.h -> td_launch.h -> td_weights.h -> td_handle.h -> the kernel headers).  The single-operator entry points and
// probes of the tests (include/tdnet_test.h, td_ops_test.h) are a SECOND library, libtdnet_hip_test.so = td_model_test.hip = this file + td_ops_test.h.
//
// Reference behaviour mirrored here (paths relative to /root/reference/Testing/model/pspnet):
//   forward / path dispatch ........ td4_psp18.py:216-229, td2_psp50.py:146-155
//   per-path graph ................. td4_psp18.py:137-212, td2_psp50.py:112-143      (td_frame.h)
//   FIFO ........................... td4_psp18.py:123-134 (depth 3), td2_psp50.py:98-109 (depth 1)
//   strict state_dict loading ...... td4_psp18.py:232-240                            (td_weights.h)
#include "td_frame.h"

extern "C" const char* tdnet_last_error(void) { return g_err; }
// The build stamps the library with a hash of its sources (tdnet_amd/build.py: -DTDNET_SRC_HASH): tests and smoke() compare it with
// the sources that were shipped, so a stale prebuilt .so cannot pass for HEAD's kernels.
#ifndef TDNET_SRC_HASH
#define TDNET_SRC_HASH "unstamped"
#endif
extern "C" const char* tdnet_version(void) { return "tdnet_amd 0.3 (gfx950, fp32 MFMA) tdnet-src-hash:" TDNET_SRC_HASH; }
// Per-handle kernel configuration (include/tdnet.h tdnet_opts); nothing here is process-wide: two handles in one process may differ.
extern "C" void tdnet_opts_default(tdnet_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->winograd = TDNET_WINOGRAD_DEFAULT;   // F(4x4,3x3) for the stride-1 3x3 convs with Cin, Cout >= 128
    o->precision = 0;                       // fp32 MFMA: the only mode the 1e-3 logits gate applies to
    o->pipeline = 1;                        // two-stage conv prefetch
    o->gemm_persistent = 1;                 // stride-1 1x1 convs and the Winograd GEMMs on the persistent multi-tile GEMM kernel
    o->attention = TDNET_ATTENTION_DEFAULT;
    o->fusion = TDNET_FUSION_DEFAULT;
    o->overlap = TDNET_OVERLAP_DEFAULT;
}
extern "C" int tdnet_create(const tdnet_cfg* cfg, tdnet_t** out) { return tdnet_create_opts(cfg, nullptr, out); }

// geometry of a handle from its configuration (the same for every handle of a weight block)
static void set_geometry(tdnet* n, const tdnet_cfg* cfg) {
    n->cfg = *cfg;
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
}

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
    TdWeights* wt = new TdWeights();
    wt->device = cfg->device;
    tdnet* n = new tdnet(wt);
    set_geometry(n, cfg);
    n->opts = opts_or_default(opts);
    n->bspec = backbone_blocks(cfg->backbone);
    build_expected(n);
    *out = n;
    return 0;
}

// A second (third, ...) handle on the SAME weights: its own workspace, K/Q/V FIFO, streams and events -- one more video stream, or one
// more frame of the same stream in flight -- while the folded, packed weights stay one copy in HBM.  The reference's batch of N samples
// shares one nn.Module's parameters the same way (td4_psp18.py:216-229).
extern "C" int tdnet_create_shared(const tdnet_t* weights_of, const tdnet_opts* opts, tdnet_t** out) {
    if (!weights_of || !out) return td_fail("tdnet_create_shared: null argument");
    if (!weights_of->finalized) return td_fail("tdnet_create_shared: the handle whose weights are to be shared is not finalized");
    if (opts) {                                                        // the packing depends on the options: a shared block has ONE set
        const tdnet_opts o = opts_or_default(opts);
        if (memcmp(&o, &weights_of->opts, sizeof(o)) != 0)
            return td_fail("tdnet_create_shared: tdnet_opts differ from those the weights were packed for (pass NULL to inherit them)");
    }
    TD_ON_DEVICE(weights_of, -1);
    tdnet* n = new tdnet(weights_of->wt);
    n->wt->refs.fetch_add(1, std::memory_order_relaxed);
    set_geometry(n, &weights_of->cfg);
    n->opts = weights_of->opts;
    if (init_handle(n)) { tdnet_destroy(n); return -1; }
    *out = n;
    return 0;
}

extern "C" void tdnet_destroy(tdnet_t* n) {
    if (!n) return;
    TD_ON_DEVICE(n);
    for (float* q : {n->img4, n->s1, n->s1b, n->bx, n->bt, n->br, n->bu, n->rowpart, n->pooled, n->ppmfeat, n->z, n->v_cur, n->q1, n->q_cur,
                     n->k1, n->vp, n->chain_a, n->chain_b, n->feat, n->ln_part, n->ln_mean, n->ln_rstd, n->ln, n->headmid,
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
    if (n->ev_cfork) hipEventDestroy(n->ev_cfork);
    if (n->ev_cjoin) hipEventDestroy(n->ev_cjoin);
    if (n->side) hipStreamDestroy(n->side);
    if (n->ev_fork) hipEventDestroy(n->ev_fork);
    if (n->ev_join) hipEventDestroy(n->ev_join);
    if (n->ev_fork2) hipEventDestroy(n->ev_fork2);
    if (n->ev_join2) hipEventDestroy(n->ev_join2);
    TdWeights* wt = n->wt;
    delete n;
    if (wt->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {                                             // the last handle of the block, whichever it is (the owner may go first)
        for (auto& p : wt->paths) free_path(p);
        delete wt;
    }
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
extern "C" int tdnet_finalize_weights(tdnet_t* n) {
    if (!n) return td_fail("tdnet_finalize_weights: null handle");
    if (n->finalized) return td_fail("tdnet_finalize_weights: already finalized");
    TD_ON_DEVICE(n, -1);
    TD_TRY(finalize_block(n));
    return init_handle(n);
}
extern "C" int tdnet_warmup(tdnet_t* n, void* stream) {
    if (!n) return td_fail("tdnet_warmup: null handle");
    if (!n->finalized || !n->ws_ready) return td_fail("tdnet_warmup: weights not finalized");
    TD_ON_DEVICE(n, -1);
    return place_chain_stream(n, (hipStream_t)stream, true);
}
extern "C" int tdnet_memory_bytes(const tdnet_t* n, size_t* weights, size_t* handle) {
    if (!n) return td_fail("tdnet_memory_bytes: null handle");
    if (weights) *weights = n->wt->device_bytes;
    if (handle) *handle = n->ws_bytes;
    return n->wt->refs.load(std::memory_order_relaxed);
}
extern "C" int tdnet_last_launch_count(const tdnet_t* n) { return n ? n->launches : -1; }
extern "C" int tdnet_streams_share_queue(void* stream_a, void* stream_b, int* shared) {
    if (!shared) return td_fail("tdnet_streams_share_queue: shared is NULL");
    *shared = 0;
    if (stream_a == stream_b) { *shared = 1; return 0; }
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) {
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        return td_fail("tdnet_streams_share_queue: hipEventCreate failed");
    }
    bool sh = false;
    const int rc = streams_share_a_queue((hipStream_t)stream_a, (hipStream_t)stream_b, e0, e1, e2, &sh);
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    *shared = sh ? 1 : 0;
    return rc;
}

// tdnet_last_launch_count: everything a frame entry enqueued, the final upsample / argmax kernel included (counted in TD_LAUNCH itself)
struct LaunchCount {
    tdnet* n; long l0;
    explicit LaunchCount(tdnet* n_) : n(n_), l0(td_launch_count) {}
    ~LaunchCount() { n->launches = (int)(td_launch_count - l0); }
};
extern "C" int tdnet_forward(tdnet_t* n, const float* img, int pos_id, float* logits, void* stream) {
    if (!n || !img || !logits) return td_fail("tdnet_forward: null argument");
    TD_ON_DEVICE(n, -1);
    LaunchCount count_(n);
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
    LaunchCount count_(n);
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
    LaunchCount count_(n);
    if (encode_frame(n, n->paths[pos_id], img, (hipStream_t)stream)) {  // a failed frame is dropped (as in tdnet_forward): no entry stays pending
        rejoin_streams(n, (hipStream_t)stream);
        n->pending_slot = n->pending_pos = -1;
        return -1;
    }
    n->pending_pos = pos_id;
    TD_HIP(hipGetLastError());
    return 0;
}
static int propagate_lowres(tdnet* n, hipStream_t s) {
    if (n->pending_slot < 0) return td_fail("tdnet_propagate: no encoded frame (call tdnet_encode first)");
    PathLayers& L = n->paths[n->pending_pos];
    const bool steady = (int)n->fifo.size() >= n->FIFO;
    if ((steady && launch_chain_now(n, L, s)) || finish_frame(n, L, steady, s)) { rejoin_streams(n, s); n->pending_slot = n->pending_pos = -1; return -1; }
    return 0;
}
extern "C" int tdnet_propagate(tdnet_t* n, float* logits, void* stream) {
    if (!n || !logits) return td_fail("tdnet_propagate: null argument");
    TD_ON_DEVICE(n, -1);
    LaunchCount count_(n);
    hipStream_t s = (hipStream_t)stream;
    if (propagate_lowres(n, s)) return -1;
    launch_upsample(n->lowres, n->cfg.nclass, n->h, n->w, n->H, n->W, logits, s);
    TD_HIP(hipGetLastError());
    return 0;
}
extern "C" int tdnet_propagate_labels(tdnet_t* n, int32_t* labels, void* stream) {
    if (!n || !labels) return td_fail("tdnet_propagate_labels: null argument");
    TD_ON_DEVICE(n, -1);
    LaunchCount count_(n);
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
    TD_HIP(hipMemcpyAsync(c.q, q, (size_t)n->Lk * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    TD_HIP(hipMemcpyAsync(c.k, k, (size_t)n->Lk * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    TD_HIP(hipMemcpyAsync(c.v, v, (size_t)n->Lk * n->DV * sizeof(float), hipMemcpyDeviceToDevice, s));
    fifo_commit(n, slot);
    return 0;
}
extern "C" int tdnet_reset(tdnet_t* n) {
    if (!n) return td_fail("tdnet_reset: null handle");
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
    else if (s == "feat") { src = n->feat_is_vcur ? n->v_cur : n->feat; C = n->DV; }   // warm-up: feat = v_cur (td4_psp18.py:142-143)
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
        TD_LAUNCH(k_ln_apply, dim3(td_grid_for((long)n->Lq * (n->DV / 4))), dim3(256), 0, (hipStream_t)0, (const float*)(n->feat_is_vcur ? n->v_cur : n->feat),
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
// which: 0 conv/GEMM kernels, 1 attention kernels, 2 everything else, 3 the dominant kernel only (128x128-tile 3x3 igemm / LDS-DMA conv /
// Winograd GEMM), 4 every 3x3 conv that reads an fp16 map (the fp16 mode's fixed roofline set).
// mode : 0 -> summed device ms, 1 -> summed algorithmic FLOP, 2 -> launch count
static double prof_query(const tdnet* n, int which, int mode) {
    if (!n || !n->prof || n->nrec == 0) return -1.0;
    double ms = 0.0, fl = 0.0, cnt = 0.0;
    int domkind = 1;                                                   // the Winograd GEMM is the dominant kernel whenever it runs
    for (size_t i = 0; i < n->nrec; ++i) if ((n->recs[i].dominant & 3) == 2) domkind = 2;
    for (size_t i = 0; i < n->nrec; ++i) {
        const ProfRec& r = n->recs[i];
        const bool take = which == 3 ? (r.family == 0 && (r.dominant & 3) == domkind) : which == 4 ? (r.family == 0 && (r.dominant & 4) != 0) : r.family == which;
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

