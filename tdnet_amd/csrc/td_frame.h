// td_frame.h -- the per-frame kernel sequence: K/Q/V FIFO, the cache-only attention chain on the side stream,
// the row-parity chains of layers 3-4, encode / finish, placement of the internal streams on hardware queues.  Part of the td_model.hip
// translation unit.
#pragma once
#include "td_launch.h"

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
    // fp16 mode: the final attention reads V' re-tiled and rounded to fp16 (td_attn_h.h).  That pass depends on V' alone: it runs HERE, on
    // the side stream under the backbone, not in front of the final attention on the critical path (round 5: one 7-us kernel + a boundary)
    if (n->vt16) {
        prof_begin(n, 2, false, 0, c);
        if (n->opts.precision >= 2) attn_prepare_vt_b3(vp, n->Lk, DV, reinterpret_cast<unsigned short*>(n->vt16), c);   // precision 2: V' re-tiled as three bf16 parts
        else attn_prepare_vt_h(vp, n->Lk, DV, n->vt16, c);
        prof_end(n, c);
    }
    TD_HIP(hipEventRecord(n->ev_join, c));
    return 0;
}
// the chain of the frame that is about to be propagated, against the FIFO as it stands
static int launch_chain_now(tdnet* n, PathLayers& L, hipStream_t s) {
    return launch_chain(n, L, s, n->vp, n->fifo[0], n->P == 4 ? n->fifo[1] : -1, n->P == 4 ? n->fifo[2] : -1);
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
    ga.wshare = 1;
    prof_begin(n, 0, 0, 2.0 * rows * W * (double)L.Cin * L.Cout, s);
    if (L.b3) gemm_b3_launch(ga, L.pers > 1 ? L.pers : 0, s);
    else gemm_launch(ga, L.tile, L.pers > 1 ? L.pers : 0, s);
    prof_end(n, s);
    return 0;
}
// (Round 5 also ran this run in the fp16 mode -- the direct fp16 LDS-DMA convs on the two row classes of their maps, two chains on two
// streams, bit-identical -- and measured 1121 -> 1050 frames/s at 720x960: a half-height conv takes exactly as long as the whole one (a
// workgroup's time is its K-loop latency chain; the chip has CUs to spare either way), the two chains run in lockstep, nothing overlaps.
// profiles/r05a_*row_parity_chains*; removed, last commit with that code: 8dff3b9.)
static int run_parity_chains(tdnet* n, PathLayers& L, int h, int w, hipStream_t s) {
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
    for (int b = sb; b < nblk; ++b) {
        BlockLayers& B = L.blocks[b];
        const float* xin = b == sb ? n->bx : n->seg_x[b - 1];
        for (int c = 0; c < 2; ++c) {
            WinoChunk ck; ck.ny = 2; ck.cy = c;
            if (!(b == sb && n->seg_conv == 1)) TD_TRY(run_wino(n, B.c1, xin, h, w, nullptr, n->seg_t[b], st[c], nullptr, ck, Vw[c], Mw[c]));
            if (B.has_ds && b > sb) TD_TRY(run_ds_rows(n, B.ds, xin, h, w, n->seg_r[b], 2, c, st[c]));
        }
        for (int c = 0; c < 2; ++c) {
            WinoChunk ck; ck.ny = 2; ck.cy = c;
            TD_TRY(run_wino(n, B.c2, n->seg_t[b], h, w, B.has_ds ? n->seg_r[b] : xin, n->seg_x[b], st[c], nullptr, ck, Vw[c], Mw[c]));
        }
    }
    TD_HIP(hipEventRecord(n->ev_cjoin, n->chain2));
    TD_HIP(hipStreamWaitEvent(s, n->ev_cjoin, 0));
    return 0;
}

static int launch_chain_now(tdnet* n, PathLayers& L, hipStream_t s);
static int encode_frame(tdnet* n, PathLayers& L, const float* img, hipStream_t s, int chain_at = -1) {   // chain_at >= 0: fork the cache-only chain in front of that backbone block
    const int DV = n->DV;
    // backbone (resnet.py:204-215)
    run_stem_pre(n, img, n->H, n->W, n->img4, s, n->opts.fusion, L.stem.stem_rows);
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
        if ((int)bi == chain_at && launch_chain_now(n, L, s)) return -1;
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
            const float* res = n->bx;
            if (B.has_ds) {                                            // conv1 and the downsample read the same map: one grouped launch where they share a kernel form
                const ConvCall both[2] = {{&B.c1, n->bx, ch, cw, n->bt}, {&B.ds, n->bx, ch, cw, n->br}};
                TD_TRY(run_conv_group(n, both, 2, s, &oh, &ow));
                res = n->br;
            } else TD_TRY(run_conv(n, B.c1, n->bx, ch, cw, nullptr, n->bt, s, &oh, &ow));
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
    if (!beside) {                                                    // one stream: the first layers side by side, then the second layers
        const ConvCall first[3] = {{&L.enc_v, n->z, n->h, n->w, n->v_cur}, {&L.enc_q0, n->z, n->h, n->w, n->q1}, {&L.enc_k0, n->z, n->h, n->w, n->k1}};
        const ConvCall second[2] = {{&L.enc_q1, n->q1, n->h, n->w, n->q_cur}, {&L.enc_k1, n->k1, n->hk, n->wk, cs.k}};
        TD_TRY(run_conv_group(n, first, 3, s));
        TD_TRY(run_conv_group(n, second, 2, s));
    } else {
        TD_TRY(run_conv(n, L.enc_q0, n->z, n->h, n->w, nullptr, n->q1, qs));
        TD_TRY(run_conv(n, L.enc_q1, n->q1, n->h, n->w, nullptr, n->q_cur, qs));
        TD_TRY(run_conv(n, L.enc_k0, n->z, n->h, n->w, nullptr, n->k1, qs));
        TD_TRY(run_conv(n, L.enc_k1, n->k1, n->hk, n->wk, nullptr, cs.k, qs));
    }
    if (!beside) {                                                    // one stream: both cache entries (q_, v_) in one launch
        prof_begin(n, 2, false, 0, s);
        TD_LAUNCH(k_subsample2, dim3(td_grid_for((long)n->Lk * (16 + DV / 4))), dim3(256), 0, s, (const float*)n->q_cur, cs.q, 64, (const float*)n->v_cur, cs.v, DV,
                  n->w, n->hk, n->wk, 4);
        prof_end(n, s);
    } else {
        prof_begin(n, 2, false, 0, qs);
        TD_LAUNCH(k_subsample, dim3(td_grid_for((long)n->Lk * 16)), dim3(256), 0, qs, (const float*)n->q_cur, cs.q, n->w, 64, n->hk, n->wk, 4);
        prof_end(n, qs);
        TD_HIP(hipEventRecord(n->ev_join2, qs));
        TD_TRY(run_conv(n, L.enc_v, n->z, n->h, n->w, nullptr, n->v_cur, s));
        prof_begin(n, 2, false, 0, s);
        TD_LAUNCH(k_subsample, dim3(td_grid_for((long)n->Lk * (DV / 4))), dim3(256), 0, s, (const float*)n->v_cur, cs.v, n->w, DV, n->hk, n->wk, 4);
        prof_end(n, s);
        TD_HIP(hipStreamWaitEvent(s, n->ev_join2, 0));
    }
    n->pending_slot = slot;
    return n->failed ? -1 : 0;
}

// chain_launched: launch_chain() already ran for this frame (it read the FIFO as it is now)
static int finish_frame(tdnet* n, PathLayers& L, bool steady, hipStream_t s) {
    const int DV = n->DV;
    const float* feat = n->v_cur;
    int stats_nstr = 0;
    n->feat_is_vcur = !steady;
    if (steady) {
        TD_HIP(hipStreamWaitEvent(s, n->ev_join, 0));                   // join: v' of the newest cached frame is ready
        const CacheSlot& ck = n->slots[n->fifo[n->FIFO - 1]];
        const AtnLayer& A = L.atn[n->P == 4 ? 2 : 0];                   // td4_psp18.py:147 / td2_psp50.py:120
        stats_nstr = (n->opts.fusion & 2) ? attn_strips(n->Lq, DV) : 0;                                         // LayerNorm strip statistics from the epilogue
        if (run_attention(n, n->q_cur, ck.k, n->vp, A.d_bias, n->v_cur, n->Lq, n->Lk, DV, n->feat, s, n->opts.attention,
                          stats_nstr ? n->ln_part : nullptr, nullptr, false, /*vt_ready=*/n->vt16 != nullptr, /*b3=*/n->opts.precision >= 2)) return -1;       // v4 + v_cur
        feat = n->feat;
    }
    // (warm-up, td4_psp18.py:142-143: feat = v_cur -- read in place; rounds 1-4 copied it into n->feat, a device copy per warm-up frame)
    // fusion bit 4: the normalised map is never written -- the head's Winograd input transform normalises while it reads `feat`
    const bool ln_in_head = (n->opts.fusion & 4) && L.head3.wino;
    const bool ln16 = L.head3.in16;                                     // fp16 mode: n->ln holds the map as fp16; the fp32 stage is made on request
    run_layernorm(n, feat, n->Lq, DV, L.d_ln_g, L.d_ln_b, n->ln_part, n->ln_mean, n->ln_rstd, ln_in_head ? nullptr : n->ln, s, stats_nstr, ln16);
    n->ln_pending = ln_in_head || ln16;
    n->ln_path = (int)(&L - &n->paths[0]);
    // fusion bit 262144 (round 6): the 1x1 classifier inside the head conv's Winograd output transform (k_wino4_out_cls): the hidden map is never
    // written, one launch fewer, low-resolution logits bit-identical to the two-kernel form
    const bool cls_in_head = (n->opts.fusion & 262144) && L.head3.wino && L.head3.chunks == 1 && wino_out_cls_supports(n->MID, n->cfg.nclass);
    const ClsArgs ca = {L.d_cls_w, L.d_cls_b, n->lowres, n->cfg.nclass};
    if (ln_in_head) {
        const LnFuse lf = {n->ln_mean, n->ln_rstd, L.d_ln_g, L.d_ln_b};
        TD_TRY(run_conv(n, L.head3, feat, n->h, n->w, nullptr, n->headmid, s, nullptr, nullptr, &lf, cls_in_head ? &ca : nullptr));
    } else
    TD_TRY(run_conv(n, L.head3, n->ln, n->h, n->w, nullptr, n->headmid, s, nullptr, nullptr, nullptr, cls_in_head ? &ca : nullptr));
    if (!cls_in_head) TD_TRY(run_classifier(n, n->headmid, n->Lq, n->MID, n->cfg.nclass, L.d_cls_w, L.d_cls_b, n->lowres, s));
    if (n->failed) return -1;
    // FIFO push (td4_psp18.py:153-154, :123-134): host bookkeeping only -- the entry's data was written by encode_frame into its slot.
    // It is the LAST thing a frame does: a frame whose head fails to launch is not in the FIFO.
    if (n->pending_slot >= 0) {
        const int slot = n->pending_slot;
        n->pending_slot = -1;
        fifo_commit(n, slot);
    }
    return 0;
}

static int frame_checks(tdnet* n, int pos_id, const char* who) {
    if (!n->finalized || !n->ws_ready) return td_fail("%s: weights not finalized (the HIP path never runs on random init)", who);
    if (pos_id < 0 || pos_id >= n->P) return td_fail("%s: pos_id %d out of range 0..%d", who, pos_id, n->P - 1);
    return 0;
}

// Error path of a frame: whatever the internal streams (cache-only attention chain, second row-parity chain) were given before the
// failure is joined back into the caller's stream, so that a failed call leaves no work of this handle running unordered behind it.
static void rejoin_streams(tdnet* n, hipStream_t s) {
    for (hipStream_t c : {n->side, n->chain2}) {
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
static int streams_share_a_queue(hipStream_t a, hipStream_t x, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, bool* shared) {
    float pair_us = 0.f;
    TD_HIP(td_streams_share_a_queue(a, x, e0, e1, e2, shared, &pair_us));          // td_device.h (the emulator's stand-in: never shared)
    if (getenv("TDNET_QUEUE_CHECK_VERBOSE")) fprintf(stderr, "tdnet queue check: spin pair %.1f us\n", pair_us);
    return 0;
}
// explicit_call: from tdnet_warmup (the documented place for the host synchronisation); otherwise the lazy check of the first frame on a
// stream tdnet_warmup has not seen -- skipped, and left for later, while that stream is being captured into a hipGraph.
static int place_chain_stream(tdnet* n, hipStream_t s, bool explicit_call = false) {
    if (!n->chain2 || std::find(n->placed_for.begin(), n->placed_for.end(), (void*)s) != n->placed_for.end()) return 0;
    if (!explicit_call && td_stream_is_capturing(s)) return 0;
    n->placed_for.push_back((void*)s);
    if (getenv("TDNET_NO_QUEUE_CHECK")) return 0;                      // A/B of this very mechanism (tools/ab_opts.py)
    // A handle alternating between caller streams checks each of them once.  chain2 is replaced only while the handle's lifetime budget of
    // replacements lasts (12 streams kept alive): a replacement that suits stream B may share a queue with stream A again, and a handle
    // that kept swapping would grow retired_streams without bound and synchronise with the host on every switch.
    if (n->placed_for.size() > 8 || n->retired_streams.size() >= 12) return 0;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    TD_HIP(hipEventCreate(&e0)); TD_HIP(hipEventCreate(&e1)); TD_HIP(hipEventCreate(&e2));
    int rc = 0;
    for (int attempt = 0; attempt < 6 && n->retired_streams.size() < 12; ++attempt) {
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
}

// started: set once the frame's own work has begun (past the argument checks) -- only then does a failure drop state
static int forward_lowres_impl(tdnet* n, const float* img, int pos_id, hipStream_t s, bool* started);
static int forward_lowres(tdnet* n, const float* img, int pos_id, hipStream_t s) {
    bool started = false;
    const int rc = forward_lowres_impl(n, img, pos_id, s, &started);
    // A frame that failed after it started is dropped: it never reaches the FIFO, and whatever the internal streams were given is joined
    // back into the caller's.  A call rejected by the checks (bad pos_id, a frame waiting for tdnet_propagate) changes nothing: the
    // pending entry of a tdnet_encode stays valid.
    if (rc && started) { rejoin_streams(n, s); n->pending_slot = n->pending_pos = -1; }
    return rc;
}
static int forward_lowres_impl(tdnet* n, const float* img, int pos_id, hipStream_t s, bool* started) {
    if (frame_checks(n, pos_id, "tdnet_forward")) return -1;
    if (n->pending_slot >= 0) return td_fail("tdnet_forward: a frame encoded with tdnet_encode is waiting for tdnet_propagate");
    PathLayers& L = n->paths[pos_id];
    n->nrec = 0;
    n->failed = false;
    TD_TRY(place_chain_stream(n, s));
    *started = true;
    const bool steady = n->cfg.model != 1 && (int)n->fifo.size() >= n->FIFO;
    // The cache-only chain overlaps the backbone: forked at the frame's start, or (fusion bit 1048576, default; fp32 and precision 2) in front of the first dilated block =
    // layer3, where it shares the chip with the large Winograd GEMMs instead of slowing the stem and layer1 (bound by the vector-memory path the chain's attention also
    // loads).  Interleaved in one process (profiles/r06al_*): td4-psp18 1024x2048 273.9 -> 275.4 (precision 2: 333.9 -> 335.4), 769x1537 392.7 -> 393.6 (450.2 -> 454.8),
    // 512x1024 838.5 -> 841.9, td2-psp50 131.5 -> 131.8; the fp16 mode LOSES (1024x2048 705.3 -> 686.7, 720x960 neutral) and keeps the early fork.  Same work, same
    // results bit for bit.  Forking in front of layer1's second block or of layer2: -1.0 % / -0.7 % with precision 2; one / two / three blocks INSIDE the row-parity run
    // (layer3's second block, layer4's blocks): 278.1 -> 275.5 / 275.3 / 273.7 fp32 (experiments TDNET_CHAIN_AT / TDNET_CHAIN_SHIFT, visits r6ak, r6am).
    int chain_at = -1;
    if (steady && (n->opts.fusion & 1048576) && n->opts.precision != 1)
        for (size_t b = 0; b < n->bspec.size() && b < L.blocks.size(); ++b)
            if (n->bspec[b].dil1 > 1 || n->bspec[b].dil2 > 1) { chain_at = (int)b; break; }
    if (steady && chain_at < 0 && launch_chain_now(n, L, s)) return -1;
    if (encode_frame(n, L, img, s, chain_at)) return -1;
    if (n->cfg.model == 1) return 0;
    return finish_frame(n, L, steady, s);
}
