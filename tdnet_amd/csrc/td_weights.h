// td_weights.h -- the weight block of a model: the strict state_dict inventory (td4_psp18.py:232-240), BN folding in fp64, packing into
// the kernels' LDS image order, upload; the row-parity plan; the per-handle workspace.  Part of the td_model.hip translation unit.
#pragma once
#include "td_handle.h"

// Upload a BN-folded OIHW weight + bias as a ConvLayer for an output of M pixels.
static int make_conv_layer(ConvLayer& L, const std::vector<float>& w, const std::vector<float>& b, int Cout, int Cin, int KS,
                           int stride, int dil, int act, bool stem, long M, const tdnet_opts& o, int forced_tile = -1, int chunks = 1) {
    L.Cin = stem ? 4 : Cin; L.Cout = Cout; L.KS = KS; L.stride = stride; L.dil = dil; L.act = act; L.stem = stem;
    L.pad = stem ? KS / 2 : dil * (KS / 2);
    L.M_out = M;
    L.rowimg_off = (o.fusion & 2048) != 0;
    L.pers = o.gemm_persistent;
    const bool deep = o.pipeline != 0;
    if (!stem && Cin % 32 != 0) return td_fail("conv: Cin=%d is not a multiple of 32", Cin);
    const bool wino_ok = o.winograd && o.precision != 1 && !stem && KS == 3 && stride == 1 && Cin % 32 == 0 && Cout % 4 == 0 &&
                         (o.winograd == 4 || (Cin >= 128 && Cout >= 128));
    L.wino = wino_ok ? 4 : 0;
    L.wino_pad = (L.wino && (o.fusion & 64) && o.gemm_persistent && gemm_supports(Cin)) ? 24 : 0;   // 24 rows: 12..48 KB between plane phases
    if (L.wino) {
        L.chunks = (chunks > 1 && dil % chunks == 0 && o.gemm_persistent && gemm_supports(Cin)) ? chunks : 1;
        chunks = L.chunks;
        L.vw = (L.chunks > 1 || (o.overlap & 2)) ? (1 << ((o.overlap >> 4) & 3)) : 0;
        L.gdma = (o.overlap & 8) != 0;
        // nb = (m+2)^2 batched [T x Cin] x [Cin x Cout] GEMMs, T = M / m^2 tiles: nb * T rows in total -> pick the tile for that many workgroups
        const int nb = (L.wino + 2) * (L.wino + 2);
        const bool pers = o.gemm_persistent && gemm_supports(Cin);
        L.tile = forced_tile >= 0 ? (ConvTile)forced_tile
               : pers ? gemm_pick_tile(wino_tiles_estimate(M, dil, L.wino) / chunks, nb, Cout, deep)
                      : conv_pick_tile((int)std::min<long>(nb * M / (L.wino * L.wino), 1 << 30), Cout, deep);
        L.CoutPad = conv_cout_pad(Cout, L.tile);
        L.nsteps = conv_nsteps(Cin, 1, 0);
        std::vector<std::vector<float>> U;
        wino_transform_weights(w.data(), Cout, Cin, L.wino, U);
        const int b3 = !(o.precision >= 2 && pers && gemm_b3_supports(Cin, Cout)) ? 0
                     : (forced_tile >= 0 || o.precision == 3) ? 1 : gemm_b3_pick(wino_tiles_estimate(M, dil, L.wino) / chunks, nb, Cout);   // a forced tile (tests, probes): always the split kernel
        if (b3) {
            // the 36 GEMMs on the bf16 MFMA, fp32-accurate (td_gemm_b3.h): the Winograd-domain weights as three bf16 parts, split here once
            L.b3 = b3;
            L.CoutPad = gemm_b3_npad(Cout);
            const size_t per = gemm_b3_packed_bytes(Cin, Cout) / 2;
            std::vector<unsigned short> packed(nb * per);
            for (int bi = 0; bi < nb; ++bi) gemm_b3_pack(U[bi].data(), Cout, Cin, packed.data() + bi * per);
            TD_TRY(dev_alloc((unsigned short**)&L.d_wp, packed.size()));
            TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        } else {
        const size_t per = (size_t)L.nsteps * 8 * L.CoutPad * 4;
        std::vector<float> packed(nb * per);
        for (int bi = 0; bi < nb; ++bi) conv_pack_weights(U[bi].data(), Cout, Cin, 1, 0, L.tile, packed.data() + bi * per);
        TD_TRY(dev_alloc(&L.d_wp, packed.size()));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        std::vector<float> bb(Cout, 0.f), zz(Cout, 0.f);
        if (!b.empty()) bb = b;
        TD_TRY(dev_alloc(&L.d_bias, (size_t)Cout));
        TD_HIP(hipMemcpy(L.d_bias, bb.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
        TD_TRY(dev_alloc(&L.d_zero, (size_t)Cout));
        TD_HIP(hipMemcpy(L.d_zero, zz.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }
    L.h16 = o.precision == 1 && !stem && Cin % 64 == 0;
    const bool stem16 = o.precision == 1 && stem && KS == 7 && stride == 2 && Cout <= 64;    // fp16-MFMA stem (td_conv_h.h)
    bool gemm1x1 = !L.h16 && !stem && KS == 1 && stride == 1 && o.gemm_persistent && gemm_supports(Cin);   // run_conv's persistent-GEMM route
    // precision 2: a stride-1 1x1 conv to <= 128 channels on a large map (a Bottleneck's conv1 in layers 1-2) is too narrow for the split GEMM's 128-column tiles
    // and would stay on the fp32 GEMM: it runs on the split DIRECT kernel instead (td_conv_ad_b3.h with KS = 1: a lane's A row is contiguous)
    // (td2-psp50 769x1537 169.2 -> 170.9 frames/s, td4-psp18 unchanged: profiles/r06ah_*)
    if (gemm1x1 && forced_tile < 0 && o.precision >= 2 && (o.fusion & 32) && (o.fusion & 524288) && Cout <= 128 && M >= 8192 && gemm_b3_supports(Cin, Cout) &&
        !gemm_b3_pick(M, 1, Cout, Cin) && conv_adirect_b3_supports(CT_128x64, Cin, KS, stem)) gemm1x1 = false;
    L.tile = forced_tile >= 0 ? (ConvTile)forced_tile : gemm1x1 ? gemm_pick_tile(M, 1, Cout, deep) : conv_pick_tile((int)M, Cout, deep);
    // precision 2: a direct conv of up to 128 output channels (a strided 3x3 / 1x1 of layer2.0, a deep stem's 64 -> 128 conv) runs as two 64-column tiles of the
    // split direct kernel (td_conv_ad_b3.h; A is loaded once per column tile) instead of the fp32 128-column kernel
    // (td2-psp50 769x1537 164.4 -> 168.0 frames/s, td4-psp18 1024x2048 351.1 -> 352.5, two processes each way on one box: profiles/r06ah_*)
    if (forced_tile < 0 && o.precision >= 2 && (o.fusion & 32) && (o.fusion & 524288) && !L.h16 && !stem && !gemm1x1 && Cout <= 128 &&
        conv_adirect_b3_supports(CT_128x64, Cin, KS, stem)) L.tile = CT_128x64;
    // fp16 mode, ResNet layer1 (64 -> 64, 3x3 stride 1): packed for the 128-wide two-wave-column tile, of which the narrow LDS-DMA kernel
    // runs the first 64-channel column (finalize_block's dma(); the second column is all padding and is never launched)
    if (L.h16 && forced_tile < 0 && !stem16 && (o.fusion & 32768) && !(o.fusion & 128) && KS == 3 && stride == 1 && Cout == 64 && Cin % 64 == 0)
        L.tile = CT_128x128_DEEP;
    L.CoutPad = conv_cout_pad(Cout, L.tile);
    if (stem16 && conv_stem_h_supports(L.tile)) {
        L.h16 = true;
        L.nsteps = conv_nsteps_stem_h();
        std::vector<_Float16> packed((size_t)L.nsteps * 8 * L.CoutPad * 8);
        conv_pack_weights_stem_h(w.data(), Cout, L.tile, packed.data());
        TD_TRY(dev_alloc((_Float16**)&L.d_wp, packed.size()));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    } else if (L.h16) {
        L.nsteps = conv_nsteps_h(Cin, KS);
        std::vector<_Float16> packed((size_t)L.nsteps * 8 * L.CoutPad * 8);
        conv_pack_weights_h(w.data(), Cout, Cin, KS, L.tile, packed.data());
        TD_TRY(dev_alloc((_Float16**)&L.d_wp, packed.size()));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    } else if (o.precision >= 2 && gemm1x1 && gemm_b3_supports(Cin, Cout) && (forced_tile >= 0 || o.precision == 3 || gemm_b3_pick(M, 1, Cout, Cin))) {
        // precision 2: a large stride-1 1x1 conv is one GEMM on the bf16 MFMA with its weights as three bf16 parts (td_gemm_b3.h)
        L.b3 = 1;
        L.CoutPad = gemm_b3_npad(Cout);
        L.nsteps = Cin / 16;
        std::vector<unsigned short> packed(gemm_b3_packed_bytes(Cin, Cout) / 2);
        gemm_b3_pack(w.data(), Cout, Cin, packed.data());
        TD_TRY(dev_alloc((unsigned short**)&L.d_wp, packed.size()));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    } else if (o.precision >= 2 && (o.fusion & 32) && (o.fusion & 524288) && !gemm1x1 && conv_adirect_b3_supports(L.tile, Cin, KS, stem)) {
        // precision 2: the Cout <= 64 convs (ResNet layer1) on the bf16 MFMA, A straight from global memory (td_conv_ad_b3.h)
        L.b3 = 1;
        L.nsteps = conv_nsteps(Cin, KS, 0);
        std::vector<unsigned short> packed(conv_adb3_packed_bytes(Cout, Cin, KS) / 2);
        conv_pack_weights_adb3(w.data(), Cout, Cin, KS, packed.data());
        TD_TRY(dev_alloc((unsigned short**)&L.d_wp, packed.size()));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    } else {
        // the 7x7 stem with its A operand straight from global memory can read a packed-row image instead of NHWC4 taps: K = 168 instead of 224
        L.stem_rows = stem && KS == 7 && stride == 2 && (o.fusion & 65536) && (o.fusion & 32) && conv_adirect_supports(L.tile, 1);
        const int stem_kind = L.stem_rows ? 2 : stem ? 1 : 0;
        L.nsteps = conv_nsteps(Cin, KS, stem_kind);
        if (L.stem_rows && o.precision >= 2 && (o.fusion & 524288)) {   // precision 2: the packed-row stem on the bf16 MFMA (td_conv_ad_b3.h STEM = 2)
            L.b3 = 1;
            std::vector<unsigned short> packed(conv_adb3_packed_bytes(Cout, Cin, KS, 2) / 2);
            conv_pack_weights_adb3(w.data(), Cout, Cin, KS, packed.data(), 2);
            TD_TRY(dev_alloc((unsigned short**)&L.d_wp, packed.size()));
            TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        } else {
        std::vector<float> packed((size_t)L.nsteps * 8 * L.CoutPad * 4);
        conv_pack_weights(w.data(), Cout, Cin, KS, stem_kind, L.tile, packed.data());
        TD_TRY(dev_alloc(&L.d_wp, packed.size()));
        TD_HIP(hipMemcpy(L.d_wp, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    L.adirect = (o.fusion & 32) && !L.h16 && !gemm1x1 && conv_adirect_supports(L.tile, 1);
    std::vector<float> bb(Cout, 0.f);
    if (!b.empty()) bb = b;
    TD_TRY(dev_alloc(&L.d_bias, (size_t)Cout));
    TD_HIP(hipMemcpy(L.d_bias, bb.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
static void free_conv_layer(ConvLayer& L) {
    if (L.d_wp) hipFree(L.d_wp);
    if (L.d_bias) hipFree(L.d_bias);
    if (L.d_zero) hipFree(L.d_zero);
    L.d_wp = L.d_bias = L.d_zero = nullptr;
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
static void free_path(PathLayers& p) {
    free_conv_layer(p.stem); free_conv_layer(p.stem2); free_conv_layer(p.stem3);
    for (auto& b : p.blocks) { free_conv_layer(b.c1); free_conv_layer(b.c2); free_conv_layer(b.c3); free_conv_layer(b.ds); }
    for (ConvLayer* c : {&p.enc_v, &p.enc_q0, &p.enc_q1, &p.enc_k0, &p.enc_k1, &p.head3}) free_conv_layer(*c);
    for (auto& a : p.atn) { free_conv_layer(a.fc); if (a.d_bias) hipFree(a.d_bias); }
    for (float* q : {p.d_ppm_w, p.d_ppm_b, p.d_ln_g, p.d_ln_b, p.d_cls_w, p.d_cls_b}) if (q) hipFree(q);
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

// Row-parity chains: which convs can run as an even-row and an odd-row half (tdnet_opts.overlap bit 1).  A stride-1 3x3 conv with an
// EVEN dilation reads, for an output row y, only the input rows y + k * dil: rows of y's parity.  So from the first such conv to the end
// of the backbone (ResNet-18/34: layer3.0.conv2 .. layer4.1.conv2, dilations 2,2,2,4,4,8,4 -- resnet.py:181-198) the even and the odd
// rows are two INDEPENDENT chains of convs (residual adds and the 1x1 downsample are pixel-wise), each with half the Winograd tiles.
constexpr int TD_CHAIN_MIN_PIXELS = 24000;                            // feature pixels (h w) from which the row-parity chains are on by default
static bool conv_chainable(int cin, int cout, int stride, int dil) {
    return stride == 1 && dil % 2 == 0 && cin >= 128 && cout >= 128 && gemm_supports(cin) && cout % 4 == 0;
}
// (fp32 Winograd convs only.  The fp16 mode's direct convs were tried as row classes too in round 5 and lost: td_frame.h.)
static void plan_chains(tdnet* n) {
    n->seg_block = -1; n->seg_conv = 0;
    const tdnet_opts& o = n->opts;
    if (!(o.overlap & 1) || o.winograd < 3 || o.precision == 1 || !o.gemm_persistent || n->deep) return;
    // The chains pay on LARGE maps only: td4-psp18, frames/s with / without them (profiles/r05g_*): 512x1024 (8192 feature pixels) 805 / 822,
    // 640x1280 (12800) 565 / 575, 769x1537 (18721) 384.5 / 391.9, 896x1792 (25088) 324.1 / 321.7, 1024x2048 (32768) 273.8 / 269.1 -- two
    // half-size GEMMs fill the chip less well than one, and below ~23 k pixels that costs more than the hidden transforms return.
    // overlap bit 4 forces them at any size (tests, A/B).
    if (n->Lq < TD_CHAIN_MIN_PIXELS && !(o.overlap & 4)) return;
    int sb = -1, sc = 0;
    for (int b = (int)n->bspec.size() - 1; b >= 0; --b) {
        const BlockSpec& S = n->bspec[b];
        if (S.bott || !conv_chainable(S.cout, S.cout, 1, S.dil2)) break;
        sb = b; sc = 1;
        if (!conv_chainable(S.cin, S.cout, S.stride, S.dil1)) break;
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
    {   // the stem's input image: NHWC4, or the packed-row image with its zero border (written once, here)
        const bool rows = !n->paths.empty() && n->paths[0].stem.stem_rows;
        const size_t img_floats = std::max((size_t)n->H * n->W * 4, rows ? (size_t)stem_rows_hp(n->H) * stem_rows_wp(n->W) * 3 + 4 : (size_t)0);
        if (dev_alloc(&n->img4, img_floats)) return -1;
        if (rows) TD_HIP(hipMemset(n->img4, 0, img_floats * sizeof(float)));
    }
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
    if (dev_alloc(&n->feat, hw * n->DV) || dev_alloc(&n->ln, hw * n->DV)) return -1;
    const size_t ln_strips = std::max<size_t>(512, (size_t)attn_strips(n->Lq, n->DV));   // k_ln_stats: <= 512 strips; attention epilogue: one per query tile
    if (dev_alloc(&n->ln_part, 2 * ln_strips * n->DV) || dev_alloc(&n->ln_mean, n->DV) || dev_alloc(&n->ln_rstd, n->DV)) return -1;
    if (n->opts.precision >= 1 && dev_alloc(&n->vt16, (size_t)(n->opts.precision == 1 ? 1 : 3) * n->DV * attn_lkpad((int)lk))) return -1;
    n->slots.resize(n->FIFO + 2);                                      // FIFO + the pending entry + one being received
    for (auto& s : n->slots)
        if (dev_alloc(&s.q, lk * 64) || dev_alloc(&s.k, lk * 64) || dev_alloc(&s.v, lk * n->DV)) return -1;
    return 0;
}
static double frame_flops(const tdnet* n);

// Builds the weight block from the host state_dict: strict inventory check, BN folding, packing, upload.  Called once per block, under
// the handle that owns the state_dict; handles created with tdnet_create_shared find the block finished.
static int finalize_block(tdnet* n) {
    AllocScope count_(&n->wt->device_bytes);
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
    n->wt->device_bytes = 0;
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
            // fp16 mode with grouped launches: the value conv on the 64-channel tile of the query / key convs it shares a launch with
            const int vtile = (n->opts.precision == 1 && (n->opts.fusion & 131072)) ? (int)conv_pick_tile((int)n->Lq, 64, n->opts.pipeline != 0) : -1;
            if (make_conv_layer(L.enc_v, fv.w, fv.b, DV, C, 1, 1, 1, 0, false, n->Lq, n->opts, vtile)) return -1;
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
    if (plan_mismatch) {
        if (attempt == 1) return td_fail("internal: conv layers ask for row-parity chunks without a chain plan");
        n->seg_block = -1; n->seg_conv = 0;                             // rebuild every layer with chunks = 1
        continue;
    }
    // precision = 1: every map between two convs of the backbone is stored as fp16 (half the conv input / output bytes; td_conv_h.h).
    // The rim: the 7x7 stem runs on the fp16 MFMA from the fp32 image and writes an fp16 map (the 3x3 deep stem's first conv stays an
    // fp32 kernel with an fp32 map), and c4 -- the LAST conv of the backbone -- writes fp32 for the pyramid, Encoding and head
    // kernels, which keep fp32 storage.
    n->act16 = n->opts.precision == 1;
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
                // ResNet layer1 (64 -> 64 channels, 3x3): one narrow tile column of the 128-wide packing (make_conv_layer packed it that way).  Isolated
                // 11.9 -> 9.8 us at 180x240, 22.0 -> 18.5 us at 256x512 against k_conv_igemm_h<128,64,..>, bit-identical (profiles/r04z_fp16_layer1_*; shipped in round 5)
                if (c.Cout == 64 && c.CoutPad == 128 && c.KS == 3 && c.stride == 1 && c.pad == c.dil && (n->opts.fusion & 32768) && conv_dma_supports(c.Cin, c.Cout, c.KS, c.tile)) {
                    c.rh = CD_128_N;
                    return;
                }
                if (c.Cout >= 128 && conv_dma_supports(c.Cin, c.Cout, c.KS, c.tile)) {
                    c.rh = conv_dma_pick_rh(c.M_out, c.Cout, c.CoutPad % 256 == 0 && !(n->opts.fusion & 1024));
                    // Small maps (720x960: 10800 output pixels): a 3x3 "same" conv with <= 256 output channels on NARROW tiles (rows x 64
                    // channels, k_conv_dma_h3n) -- half the weight bytes per K step and CU, the term that dominates there: 128 channels
                    // 13.8 -> 10.3 us, 256 channels 20.6 -> 19.8 us isolated (profiles/r04u_*).  No gain at 32768 pixels.
                    const bool same3 = c.KS == 3 && c.stride == 1 && c.pad == c.dil;
                    if ((n->opts.fusion & 32768) && same3 && c.M_out <= 16384 && c.Cout <= 256)
                        c.rh = c.Cout <= 128 ? CD_128_N : CD_192_N;       // (256 channels on 128 x 64 tiles as well: 2.1 % instead of 2.6 % in the frame;
                                                                          //  512 channels, 64- and 96-row tiles: slower, profiles/r05c_*)
                    else if (n->opts.fusion & 8192)                     // the 128- and 192-row tiles with four dedicated loader waves (k_conv_dma_h3p):
                        c.rh = c.rh == CD_128 ? CD_128_P : c.rh == CD_192 ? CD_192_P : c.rh;
                        // isolated 22.5 -> 21.0 / 57.8 -> 56.7 us; 256 rows: no gain (profiles/r04d_*).  (192 rows with TWELVE matrix waves of 32 x 64 -- three
                        // per SIMD instead of two SIMDs with twice the MFMAs -- is 3-5 % faster alone and 0.6 % SLOWER in the frame: profiles/r04z_*.)
                }
            };
            if (n->deep) dma(L.stem3);
            for (auto& B : L.blocks) { dma(B.c1); dma(B.c2); if (B.bott) dma(B.c3); if (B.has_ds) dma(B.ds); }
            // The head's 3x3 conv (d_v -> d_v / 4 channels; >= 128 for td4): LayerNorm writes its map as fp16 -- the rounding the conv
            // applied to the fp32 map while staging it -- and the conv runs on the LDS-DMA kernel (1024x2048: 77 -> 46 us).
            if (n->cfg.model != 1 && L.head3.h16 && !L.head3.wino) { L.head3.in16 = true; dma(L.head3); if (!L.head3.rh) L.head3.in16 = false; }
        }
    break;
    }
    n->sd.clear();
    n->finalized = true;
    n->flops_frame = frame_flops(n);
    return 0;
}

// The per-handle half: workspace, K/Q/V FIFO slots, internal streams and events.  Needs a finished weight block (layer geometry).
static int init_handle(tdnet* n) {
    if (n->ws_ready) return 0;
    {   // A frame runs on three hardware queues at once; with HIP's default of 4 queues per priority class a process that creates a few
        // streams of its own (torch's pools, RCCL) owns enough queues that they are no longer all resident, and frames run at 0.66x
        // (tdnet_amd/__init__.py, profiles/r04l_*).  bench.py / tests / __graft_entry__ export GPU_MAX_HW_QUEUES=2 before the runtime starts; any other caller is told once.
        static std::atomic<bool> told{false};                         // once per process, whichever thread gets there first
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (!getenv("TDNET_QUIET") && (!q || atoi(q) > 3) && !told.exchange(true)) {
            fprintf(stderr, "tdnet: GPU_MAX_HW_QUEUES is %s; export GPU_MAX_HW_QUEUES=2 before the HIP runtime starts -- with more hardware queues "
                            "alive in the process this handle's streams are time-sliced (275 -> 185 frames/s measured behind an RCCL communicator)\n", q ? q : "not set");
        }
    }
    {
        AllocScope count_(&n->ws_bytes);
        if (alloc_workspace(n)) return -1;
    }
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
    if (n->seg_block >= 0) {
        // (Which hardware queue this stream gets is HIP's choice; place_chain_stream() checks it against the caller's in tdnet_warmup / at the first frame.)
        TD_HIP(hipStreamCreateWithFlags(&n->chain2, hipStreamNonBlocking));
        TD_HIP(hipEventCreateWithFlags(&n->ev_cfork, hipEventDisableTiming));
        TD_HIP(hipEventCreateWithFlags(&n->ev_cjoin, hipEventDisableTiming));
    }
    TD_HIP(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
    TD_HIP(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
    TD_HIP(hipEventCreateWithFlags(&n->ev_fork2, hipEventDisableTiming));
    TD_HIP(hipEventCreateWithFlags(&n->ev_join2, hipEventDisableTiming));
    n->ws_ready = true;
    return 0;
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
