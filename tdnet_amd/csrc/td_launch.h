// td_launch.h -- one host-side launch helper per operator of the frame (conv / Winograd conv / attention / plane LayerNorm / pyramid slice /
// stem / classifier / upsample) + the per-launch profiling records.  Part of the td_model.hip translation unit.
#pragma once
#include "td_weights.h"

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

// Winograd F(4x4) conv (or one chunk of it): input transform -> 36 batched GEMMs -> output transform, all on stream s.
// V / Mb: workspaces for THIS call ([nb][Tc + pad][C]); nullptr = the handle's (n->wino_v / wino_m) or, without a handle, temporary ones.
// cls != nullptr: the conv is the FCN head's 3x3 and its output transform also applies the 1x1 classifier (k_wino4_out_cls): `out` is not written
static int run_wino(tdnet* n, const ConvLayer& L, const float* in, int H, int W, const float* resid, float* out, hipStream_t s,
                    const LnFuse* lnf, const WinoChunk& ck, float* V, float* Mb, const ClsArgs* cls = nullptr) {
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
        if (out_side && cls) {
            const unsigned grid = (unsigned)((Tc + 3) / 4);
            if (L.Cout == 128) TD_LAUNCH((k_wino4_out_cls<2>), dim3(grid), dim3(256), wino_out_cls_lds(128, cls->NC), s, wa, *cls);
            else TD_LAUNCH((k_wino4_out_cls<1>), dim3(grid), dim3(256), wino_out_cls_lds(64, cls->NC), s, wa, *cls);
        } else
        if (L.vw == 1) launch_wino4_c<1>(out_side, wa, s);
        else if (L.vw == 2 && C % 2 == 0) launch_wino4_c<2>(out_side, wa, s);
        else if (L.vw == 4 && C % 4 == 0) launch_wino4_c<4>(out_side, wa, s);
        else if (L.vw) launch_wino4_c<1>(out_side, wa, s);
        else if (out_side) TD_LAUNCH(k_wino4_out, dim3(td_grid_for(T * (L.Cout / 4), 256, 256 * 16)), dim3(256), 0, s, wa);
        else TD_LAUNCH(k_wino4_in, dim3(td_grid_for(T * (L.Cin / 4), 256, 256 * 16)), dim3(256), 0, s, wa);
        prof_end(n, s);
    };
    transform(false);
    prof_begin(n, 0, 2, 2.0 * nb * Tc * (double)L.Cin * L.Cout, s);
    if (n && (probe_skip() & 8)) { /* timing probe: no GEMMs */ }
    else if (L.pers && gemm_supports(L.Cin)) {
        GemmArgs ga;
        ga.a = V; ga.wp = L.d_wp; ga.bias = L.d_zero; ga.resid = nullptr; ga.out = Mb;
        ga.M = (int)Tc; ga.N = L.Cout; ga.NPad = L.CoutPad; ga.K = L.Cin; ga.nbatch = nb; ga.act = 0; ga.tiles_m = ga.tiles_n = 0; ga.MP = (int)TP;
        // the split GEMM's persistent grid: 512 workgroups (two per CU); a row-parity CHUNK's GEMM -- two of them are in flight, one per chain -- takes 320: of
        // layer 4's 576 tiles 256 workgroups then walk two and the sibling's first workgroups find a free slot at once.  Frame with precision 2 at 1024x2048, by grid of the
        // chunks' GEMMs (profiles/r06ao_*): 512: 339.4, 288: 338.6, 320: 346.0, 352: 344.7, 384: 344.5, 448: 342.1 frames/s.
        if (L.b3) gemm_b3_launch(ga, L.pers > 1 ? L.pers : chunked ? 320 : 0, s);
        // (the fp32 kernel keeps its 768 workgroups for the chunks too: 448 / 512 / 576 / 640 / 1024 gave 273.1 / 273.4 / 271.8 / 275.5 / 274.3 against 278.9 frames/s)
        else if (L.gdma && gemm_dma_supports(L.Cin, L.Cout, L.tile)) gemm_dma_launch(ga, L.pers > 1 ? L.pers : 0, s);
        else gemm_launch(ga, L.tile, L.pers > 1 ? L.pers : 0, s);
    } else {
        ConvArgs g;
        g.in = V; g.wp = L.d_wp; g.bias = L.d_zero; g.resid = nullptr; g.out = Mb;
        g.H = 1; g.W = (int)Tc; g.Cin = L.Cin; g.Wo = (int)Tc; g.Cout = L.Cout; g.CoutPad = L.CoutPad;
        g.stride = 1; g.dil = 1; g.pad = 0; g.M = (int)Tc; g.nsteps = L.nsteps; g.act = 0; g.tiles_n = 0; g.nbatch = nb;
        conv_launch(g, L.tile, 1, false, s);
    }
    prof_end(n, s);
    transform(true);
    if (own) { TD_HIP(hipStreamSynchronize(s)); hipFree(V); hipFree(Mb); }
    return 0;
}

// The fp16 LDS-DMA conv of layer L on the arguments a: a cascade of kernel forms for the SAME tile, each falling back to the next when
// the conv does not qualify (not a 3x3 "same" conv, halo wider than the form's image buffer): narrow tiles -> loader waves -> row
// images (one LDS image per kernel ROW, k_conv_dma_h3) -> tap by tap.
static void launch_conv_dma_forms(const ConvLayer& L, const ConvArgs& a, hipStream_t s) {
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

// out[Ho*Wo][Cout] = act(conv(in[H][W][Cin]) + bias (+ resid))
static int run_conv(tdnet* n, const ConvLayer& L, const float* in, int H, int W, const float* resid, float* out, hipStream_t s,
                    int* Ho_out = nullptr, int* Wo_out = nullptr, const LnFuse* lnf = nullptr, const ClsArgs* cls = nullptr) {
    if (lnf && !L.wino) return td_fail("internal: LayerNorm fusion needs a Winograd input transform");
    if (cls && (!L.wino || L.chunks > 1 || resid)) return td_fail("internal: the classifier rides in a whole Winograd conv's output transform");
    const int Ho = out_size(H, L.KS, L.stride, L.dil, L.pad), Wo = out_size(W, L.KS, L.stride, L.dil, L.pad);
    if (L.wino) {
        if (Ho_out) *Ho_out = H;
        if (Wo_out) *Wo_out = W;
        if (L.chunks > 1 && (!n || !n->ws_ready)) {                    // operator tests and probes: the chunks one after the other on one stream
            for (int c = 0; c < L.chunks; ++c) {
                WinoChunk ck; ck.ny = L.chunks; ck.cy = c;
                TD_TRY(run_wino(n, L, in, H, W, resid, out, s, lnf, ck, nullptr, nullptr));
            }
            return 0;
        }
        return run_wino(n, L, in, H, W, resid, out, s, lnf, WinoChunk(), nullptr, nullptr, cls);
    }
    ConvArgs a;
    a.in = in; a.wp = L.d_wp; a.bias = L.d_bias; a.resid = resid; a.out = out;
    a.H = H; a.W = W; a.Cin = L.Cin; a.Wo = Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.stride = L.stride; a.dil = L.dil; a.pad = L.pad; a.M = Ho * Wo; a.nsteps = L.nsteps; a.act = L.act; a.tiles_n = 0; a.nbatch = 1;
    // dominant: bits 0-1 = 1 for the 128x128-tile / LDS-DMA 3x3 convs (the kernel set of rounds 3-4), bit 2 = EVERY 3x3 conv that reads an fp16 map
    // (the fixed set of the fp16 mode's roofline since round 5: routing a layer to another kernel does not change it)
    prof_begin(n, 0, (((L.tile == CT_128x128 || L.tile == CT_128x128_DEEP || L.rh) && L.KS == 3 && !L.stem) ? 1 : 0) | ((L.h16 && L.in16 && L.KS == 3 && !L.stem) ? 4 : 0), L.flops_per_pixel() * a.M, s);
    if (L.h16 && L.stem) conv_launch_stem_h(a, L.out16, s);
    else if (L.h16 && L.rh) launch_conv_dma_forms(L, a, s);
    else if (L.h16) conv_launch_h(a, L.tile, L.KS, L.in16, L.out16, s);
    else if (L.adirect && L.b3 && !L.stem_rows) conv_launch_adirect_b3(a, L.KS, 0, s);   // precision 2, narrow convs (incl. the stride-1 1x1 ones kept off the GEMM route)
    else if (L.pers && L.KS == 1 && L.stride == 1 && !L.stem && gemm_supports(L.Cin)) {
        GemmArgs ga;
        ga.a = in; ga.wp = L.d_wp; ga.bias = L.d_bias; ga.resid = resid; ga.out = out;
        ga.M = a.M; ga.N = L.Cout; ga.NPad = L.CoutPad; ga.K = L.Cin; ga.nbatch = 1; ga.act = L.act; ga.tiles_m = ga.tiles_n = 0; ga.MP = a.M;
        if (L.b3) gemm_b3_launch(ga, L.pers > 1 ? L.pers : 0, s);
        else gemm_launch(ga, L.tile, L.pers > 1 ? L.pers : 0, s);
    } else if (L.adirect && L.stem_rows) {                              // the image is padded already: its own geometry, 3 channels, no padding taps
        ConvArgs r = a;
        r.H = stem_rows_hp(H); r.W = stem_rows_wp(W); r.Cin = 3; r.pad = 0;
        if (L.b3) conv_launch_adirect_b3(r, L.KS, 2, s);
        else conv_launch_adirect(r, L.KS, 2, s);
    }
    else if (L.adirect) conv_launch_adirect(a, L.KS, L.stem ? 1 : 0, s);
    else conv_launch(a, L.tile, L.KS, L.stem, s);
    prof_end(n, s);
    if (Ho_out) *Ho_out = Ho;
    if (Wo_out) *Wo_out = Wo;
    return 0;
}

struct ConvCall { const ConvLayer* L; const float* in; int H, W; float* out; };
// Up to three independent convs in one launch when they share a kernel form (fp16 mode: the register-staged k_conv_igemm_h on the same tile,
// kernel size and storage types; `fusion` bit 131072); otherwise one launch each, in order.
static int run_conv_group(tdnet* n, const ConvCall* c, int ng, hipStream_t s, int* Ho_out = nullptr, int* Wo_out = nullptr) {   // Ho / Wo: of c[0]
    bool same = n && (n->opts.fusion & 131072) && ng >= 2 && ng <= 3;
    for (int g = 0; same && g < ng; ++g) {
        const ConvLayer& L = *c[g].L;
        const ConvLayer& L0 = *c[0].L;
        same = L.h16 && !L.stem && !L.rh && !L.wino && L.in16 == L.out16 && L.in16 == L0.in16 && (g == 0 ? (L.KS == 1 || L.KS == 3) : L.KS == 1) &&
               conv_tile_dims(L.tile).BM == conv_tile_dims(L0.tile).BM && conv_tile_dims(L.tile).BN == conv_tile_dims(L0.tile).BN;
    }
    if (!same) {
        for (int g = 0; g < ng; ++g) TD_TRY(run_conv(n, *c[g].L, c[g].in, c[g].H, c[g].W, nullptr, c[g].out, s, g == 0 ? Ho_out : nullptr, g == 0 ? Wo_out : nullptr));
        return 0;
    }
    ConvArgs a[3];
    double flops = 0.0;
    for (int g = 0; g < ng; ++g) {
        const ConvLayer& L = *c[g].L;
        const int Ho = out_size(c[g].H, L.KS, L.stride, L.dil, L.pad), Wo = out_size(c[g].W, L.KS, L.stride, L.dil, L.pad);
        a[g].in = c[g].in; a[g].wp = L.d_wp; a[g].bias = L.d_bias; a[g].resid = nullptr; a[g].out = c[g].out;
        a[g].H = c[g].H; a[g].W = c[g].W; a[g].Cin = L.Cin; a[g].Wo = Wo; a[g].Cout = L.Cout; a[g].CoutPad = L.CoutPad;
        a[g].stride = L.stride; a[g].dil = L.dil; a[g].pad = L.pad; a[g].M = Ho * Wo; a[g].nsteps = L.nsteps; a[g].act = L.act; a[g].tiles_n = 0; a[g].nbatch = 1;
        flops += L.flops_per_pixel() * a[g].M;
        if (g == 0 && Ho_out) *Ho_out = Ho;
        if (g == 0 && Wo_out) *Wo_out = Wo;
    }
    // a 3x3 conv on an fp16 map stays in the fp16 roofline's fixed layer set: the record carries ITS FLOP only (the launch's time includes the 1x1 beside it)
    const bool dom = c[0].L->in16 && c[0].L->KS == 3;
    prof_begin(n, 0, dom ? 4 : 0, dom ? c[0].L->flops_per_pixel() * a[0].M : flops, s);
    conv_launch_h_group(a, ng, c[0].L->tile, c[0].L->KS, c[0].L->in16, c[0].L->out16, s);
    prof_end(n, s);
    return 0;
}

// ln_part != nullptr: the kernel also writes the plane-LayerNorm strip statistics of `out` (one strip per 32-row query tile)
// vt16 (or the handle's n->vt16): the operand buffer of the fp16-MFMA kernel (tdnet_opts.precision = 1) or, with b3, of the split kernel (precision 2:
// td_attn_b3.h; the handle uses it for the FINAL attention of a frame only -- the cached-frame steps are hidden on the side stream either way)
static int run_attention(tdnet* n, const float* q, const float* k, const float* vp, const float* bias, const float* resid,
                         int Lq, int Lk, int DV, float* out, hipStream_t s, int online = 0, float* ln_part = nullptr,
                         _Float16* vt16 = nullptr, bool slices = false, bool vt_ready = false, int b3 = 0) {   // b3: 1 = the split kernels (form by size), 2 / 3 = the 32- / 64-query form (tests)
    if (n && n->vt16 && (n->opts.precision == 1 || b3)) vt16 = n->vt16;
    AttnArgs a;
    a.q = q; a.k = k; a.vp = vp; a.bias = bias; a.resid = resid; a.out = out; a.Lq = Lq; a.Lk = Lk;
    a.scale_log2e = 1.4426950408889634f / 8.0f;                        // temperature = sqrt(d_k) = 8 (transformer.py:65)
    a.ln_part = ln_part; a.ln_nstr = 0;
    if (n && (probe_skip() & 4) && Lq > Lk) return 0;
    prof_begin(n, 1, false, 2.0 * Lq * (double)Lk * (64 + DV), s);
    const int rc = (vt16 && b3) ? attn_launch_b3(a, DV, reinterpret_cast<unsigned short*>(vt16), s, vt_ready, b3 - 1)
                 : vt16 ? attn_launch_h(a, DV, vt16, s, vt_ready) : attn_launch(a, DV, online, s, slices);   // vt16: the fp16-MFMA kernel (tdnet_opts.precision = 1)
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
    (void)pooled;
    TD_LAUNCH(k_ppm_rowbins, dim3(h * ((C + 511) / 512)), dim3(1024), at.n * 512 * 4, s, c4, rowpart, w, C, at);   // rowpart: [h][12][C] row bins
    TD_LAUNCH(k_ppm_pool_conv, dim3(50 * (FS / 64)), dim3(256), (256 + C) * 4, s, (const float*)rowpart, wgt, bias, ppmfeat, h, w, C, FS);
    TD_LAUNCH(k_ppm_assemble, dim3(td_grid_for((long)h * w * (C / 4))), dim3(256), 0, s, c4, (const float*)ppmfeat, z, h, w, C,
              pid * XS, XS, FS);
    prof_end(n, s);
}

// rows: the packed-row image of the 7x7 stem (ConvLayer.stem_rows; img4 then holds [H + 7][W + 8][3] with a zero border) instead of NHWC4
static void run_stem_pre(tdnet* n, const float* img, int H, int W, float* img4, hipStream_t s, int fusion, bool rows = false) {
    prof_begin(n, 2, false, 0, s);
    if (rows)
        TD_LAUNCH(k_nchw3_to_rgbpad, dim3(td_grid_for((long)H * ((W + 3) / 4))), dim3(256), 0, s, img, img4, H, W, stem_rows_wp(W));
    else if ((fusion & (16 | 256)) && (H * W) % 4 == 0 && ((size_t)img & 15) == 0)
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
    if (W % 4 == 0 && ((size_t)out & 15) == 0 && H <= 65535 && C <= 65535) TD_LAUNCH(k_upsample_x4, dim3((W / 4 + 255) / 256, H, C), dim3(256), 0, s, in, out, C, h, w, H, W);
    else if (H <= 65535 && C <= 65535) TD_LAUNCH(k_upsample_row, dim3((W / 4 + 2 + 255) / 256, H, C), dim3(256), 0, s, in, out, C, h, w, H, W);
    else TD_LAUNCH(k_upsample, dim3(td_grid_for((long)C * H * W, 256, 256 * 16)), dim3(256), 0, s, in, out, C, h, w, H, W);
}
