"""CPU-side verification of the HIP kernel SOURCES through the test-only fiber emulator (tests/emu/): every kernel
family on ragged shapes, then the whole per-frame pipeline against the golden vectors captured from the reference.
These run without a GPU; the GPU parity tests proper are tests/test_gpu_*.py."""
import os

import numpy as np
import pytest

import emu_util
import opcheck
from tdnet_amd import _capi, arch, weights
from tdnet_amd.engine import Engine

MEM = opcheck.NumpyMem()


@pytest.fixture(scope="module")
def lib():
    return emu_util.emu_lib()


DIRECT = {"winograd": 0}     # force the direct implicit-GEMM kernels (the library default routes wide stride-1 3x3 convs to Winograd)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_conv_variants(lib, tile):
    opcheck.conv(lib, MEM, 13, 21, 64, 128, 3, 1, 2, 1, True, tile, opts=DIRECT)       # dilated 3x3 + residual + ReLU
    opcheck.conv(lib, MEM, 13, 21, 32, 96, 3, 2, 1, 0, False, tile, opts=DIRECT)       # stride 2, Cout not a tile multiple
    opcheck.conv(lib, MEM, 11, 19, 64, 19, 1, 1, 1, 2, False, tile, opts=DIRECT)       # 1x1, 19 channels, LeakyReLU
    opcheck.conv(lib, MEM, 17, 9, 128, 64, 1, 2, 1, 0, True, tile, opts=DIRECT)        # 1x1 stride-2 downsample
    opcheck.conv(lib, MEM, 12, 30, 64, 160, 3, 1, 4, 1, False, tile, opts=DIRECT)      # dilation 4, two N tiles
    opcheck.conv(lib, MEM, 7, 9, 32, 64, 1, 1, 1, 0, False, tile, opts=DIRECT)         # a single K step (pipeline prologue only)
    opcheck.conv(lib, MEM, 7, 9, 64, 64, 1, 1, 1, 0, True, tile, opts=DIRECT)          # two K steps
    opcheck.conv(lib, MEM, 7, 9, 96, 64, 1, 1, 1, 1, False, tile, opts=DIRECT)         # three K steps (odd tail of the 2-stage loop)


def test_conv_auto_tile_and_edges(lib):
    opcheck.conv(lib, MEM, 20, 23, 64, 64, 1, 4, 1, 2, False, opts=DIRECT)             # the stride-4 key sub-sampling conv
    opcheck.conv(lib, MEM, 9, 17, 128, 256, 3, 1, 8, 1, True, opts=DIRECT)             # dilation 8 larger than the image half
    opcheck.conv(lib, MEM, 5, 9, 256, 512, 3, 1, 16, 1, False, opts=DIRECT)            # dilation 16 (resnet34 multi-grid): all taps but centre padded
    opcheck.conv(lib, MEM, 40, 40, 32, 128, 3, 1, 1, 1, False, opts=DIRECT)            # several M tiles, ragged last tile
    opcheck.conv(lib, MEM, 1, 1, 32, 32, 3, 1, 1, 0, False, opts=DIRECT)               # single pixel


def test_conv_a_operand_direct_from_global(lib):
    """td_conv_ad.h (fusion bit 32): the Cout <= 64 convs with the A operand loaded straight from global memory in fragment layout:
    3x3 with dilation / stride / padding on every side, 1x1 with stride, ragged M, 1..3 K steps, residual and activations."""
    o = {"winograd": 0, "fusion": 32}
    for tile in (2, 5, None):
        opcheck.conv(lib, MEM, 13, 21, 64, 64, 3, 1, 1, 1, True, tile, opts=o)
        opcheck.conv(lib, MEM, 13, 21, 64, 64, 3, 1, 2, 0, False, tile, opts=o)
        opcheck.conv(lib, MEM, 12, 17, 32, 48, 3, 2, 1, 2, True, tile, opts=o)
        opcheck.conv(lib, MEM, 17, 9, 128, 64, 1, 2, 1, 0, True, tile, opts=o)
        opcheck.conv(lib, MEM, 7, 9, 96, 19, 1, 4, 1, 1, False, tile, opts=o)
        opcheck.conv(lib, MEM, 40, 40, 32, 64, 3, 1, 4, 1, False, tile, opts=o)
        opcheck.conv(lib, MEM, 1, 1, 32, 32, 3, 1, 1, 0, False, tile, opts=o)
    for (H, W) in ((33, 65), (40, 52), (8, 10)):
        opcheck.stem(lib, MEM, H, W, opts={"fusion": 32})


def test_stem(lib):
    opcheck.stem(lib, MEM, 33, 65)
    opcheck.stem(lib, MEM, 40, 52)
    for (H, W) in ((40, 52), (33, 65), (34, 66), (8, 10)):                # fusion bit 16: 4-pixel layout kernel (H*W % 4 == 0) and 2-output max-pool, odd and even widths
        opcheck.stem(lib, MEM, H, W, opts={"fusion": 16})
    # round 5: the 7x7 stem on the PACKED-ROW image (fusion bit 65536 with bit 32; [H + 7][W + 8][3] with a zero border, a K step = one kernel
    # row of 21 contiguous floats, K = 168): odd and even sizes, images narrower than one 24-float row read, a single output row
    for (H, W) in ((33, 65), (40, 52), (34, 66), (8, 10), (9, 9), (129, 17), (7, 31)):
        opcheck.stem(lib, MEM, H, W, opts={"fusion": 32 | 65536})


def test_attention(lib):
    opcheck.attention(lib, MEM, 45, 6, 512)                                # Lk smaller than one key tile
    opcheck.attention(lib, MEM, 153, 15, 512, False, False)
    opcheck.attention(lib, MEM, 300, 200, 512, spike=True)                 # ragged super-tiles + a dominating key
    opcheck.attention(lib, MEM, 45, 6, 128)
    opcheck.attention(lib, MEM, 200, 131, 128, True, False, qk_scale=2.0)
    opcheck.attention(lib, MEM, 64, 128, 512)
    opcheck.attention(lib, MEM, 1, 1, 128)
    opcheck.attention(lib, MEM, 130, 193, 128, True, True, spike=True)     # d_v 128 variant: two query tiles x two channel halves, ragged both ways
    opcheck.attention(lib, MEM, 64, 64, 128, False, True)


def test_attention_channel_slices(lib):
    """td_attn.h: DV = 512 as two 256-channel slices in one launch (grid.y = 2, k_attention<1,4,2,*>: the cached-frame steps of td4's
    propagation chain) for the three softmax schedules, ragged query / key tiles, bias and residual through the sliced pointers."""
    for online in (0, 1, 2):
        opcheck.attention(lib, MEM, 45, 6, 512, online=online | 64)
        opcheck.attention(lib, MEM, 153, 200, 512, True, True, spike=True, online=online | 64)
        opcheck.attention(lib, MEM, 64, 128, 512, False, False, online=online | 64)


def test_attention_online_softmax_and_layernorm_statistics(lib):
    """tdnet_opts.attention = 1 (single pass, lazily moved reference) and fusion bit 2 (plane-LayerNorm strip statistics written by
    the epilogue): same gate as the two-pass kernel, on ragged shapes, with a dominating key, and with scores that keep growing
    along the key axis (the reference has to move several times per query tile)."""
    for online in (1, 2, 0):
        opcheck.attention(lib, MEM, 45, 6, 512, online=online, ln=True)
        opcheck.attention(lib, MEM, 300, 200, 512, spike=True, online=online, ln=True)
        opcheck.attention(lib, MEM, 130, 193, 128, True, True, spike=True, online=online, ln=True)
        opcheck.attention(lib, MEM, 33, 1, 128, online=online, ln=True)                  # a second strip with one row; a single key
        opcheck.attention(lib, MEM, 97, 300, 512, ramp=True, online=online, ln=True)
        opcheck.attention(lib, MEM, 70, 260, 128, False, True, ramp=True, online=online, ln=True)   # (without the residual the plane is nearly constant: LayerNorm of it is ill-conditioned)
    opcheck.attention(lib, MEM, 200, 131, 128, True, False, qk_scale=2.0, online=1)
    opcheck.attention(lib, MEM, 1, 1, 128, online=1)
    opcheck.attention(lib, MEM, 64, 128, 512, online=2)                                  # exactly one super-tile
    opcheck.attention(lib, MEM, 40, 129, 128, online=2)                                  # two full super-tiles + one key


def test_layernorm_ppm_upsample(lib):
    for hw, c in [(45, 512), (153, 128), (1000, 512), (2145, 512), (1, 128), (1300, 256)]:   # 2145: empty strips (512 strips of 5 pixels)
        opcheck.layernorm(lib, MEM, hw, c)
    for h, w, pid in [(5, 9, 0), (9, 17, 1), (13, 25, 1), (6, 6, 0), (97 // 4, 193 // 4, 0), (7, 11, 1), (4, 193, 0)]:   # row sums by atoms: odd widths
        opcheck.ppm(lib, MEM, h, w, pid)
    for c, h, w, H, W in [(19, 5, 9, 33, 65), (3, 9, 17, 65, 129), (2, 1, 1, 4, 5)]:
        opcheck.upsample(lib, MEM, c, h, w, H, W)
    # k_upsample_row (round 5; any W -- 769x1537 is the reference's native size): rows of every alignment, a destination that itself starts
    # at every 4-byte offset of a 16-byte line (sample i > 0 of a batch), widths shorter than a quad, nothing written outside the map;
    # against the 4-column kernel's answer where W % 4 == 0 allows one
    import torch
    import torch.nn.functional as F
    for c, h, w, H, W in [(3, 5, 9, 33, 65), (2, 4, 7, 29, 53), (2, 3, 2, 9, 3), (1, 2, 2, 5, 1), (2, 5, 9, 33, 66), (2, 5, 9, 33, 64)]:
        x = np.random.default_rng(W).standard_normal((c, h, w)).astype(np.float32)
        ref = F.interpolate(torch.from_numpy(x)[None], (H, W), mode="bilinear", align_corners=True)[0].numpy()
        outs = []
        for off in range(4):
            buf = np.full(c * H * W + 16, 7e7, np.float32)
            base = (-buf.ctypes.data // 4) % 4                                   # element index of a 16-byte boundary
            view = buf[base + off: base + off + c * H * W]
            lib.check(lib.tdnet_op_upsample(MEM.ptr(MEM.put(x)), c, h, w, H, W, view.ctypes.data, None))
            assert np.abs(view.reshape(c, H, W) - ref).max() <= 1e-5, (c, h, w, H, W, off)
            assert (buf[:base + off] == 7e7).all() and (buf[base + off + c * H * W:] == 7e7).all(), (W, off)
            outs.append(view.copy())
        assert all(np.array_equal(outs[0], o) for o in outs[1:]), (W, "the value of an element must not depend on which store wrote it")


CASES = [("td4", "resnet18", 33, 65), ("td2", "resnet18", 33, 65), ("td2", "resnet50", 33, 65), ("td2", "resnet34", 33, 65), ("td2", "resnet18", 49, 81),
         ("td4", "resnet34", 33, 65), ("td4", "resnet50", 33, 65), ("td4", "resnet18", 65, 129)]


@pytest.mark.parametrize("name,bb,H,W", CASES)
def test_full_pipeline_against_reference_goldens(lib, golden_dir, name, bb, H, W):
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    hk, wk = arch.key_size(h), arch.key_size(w)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    T = min(7 if name == "td4" else 2, 1 + max(int(k.split("_")[0][1:]) for k in g.files if k.startswith("f")))   # td4: t = 6 is forward_path3 in steady state
    e = Engine(spec.path_num, int(bb[6:]), 19, H, W, 0, lib=lib)
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    shapes = {"c4": (1, spec.d_model, h, w), "z": (1, spec.d_model, h, w), "v_cur": (1, spec.d_v, h, w), "q_cur": (1, h * w, 64),
              "ln": (1, spec.d_v, h, w), "lowres": (1, 19, h, w), "cache_q": (1, hk * wk, 64), "cache_k": (1, hk * wk, 64),
              "cache_v": (1, hk * wk, spec.d_v)}
    for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
        out = np.full((1, 19, H, W), 7e7, np.float32)
        e.forward(x, t % spec.path_num, out)
        assert e.fifo_len() == min(t + 1, spec.fifo)
        for st, shp in shapes.items():
            if "f%d_%s" % (t, st) not in g.files:                               # the larger goldens keep logits and a few stages only
                continue
            ref = g["f%d_%s" % (t, st)]
            got = e.stage(st, shp)
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (t, st)
        ref = g["f%d_logits" % t]
        err = float(np.abs(out - ref).max())
        assert err <= 1e-3                                                       # north_star: logits within 1e-3 fp32
        bad = out[0].argmax(0) != ref[0].argmax(0)
        if bad.any():                                                            # label flips only inside the reference's top-2 tie band
            top2 = np.sort(ref[0], axis=0)[-2:]
            assert ((top2[1] - top2[0])[bad] <= 2 * err).all(), (t, int(bad.sum()))
        lab = np.zeros((H, W), np.int32)
        e.argmax(out, lab)
        assert (lab == out[0].argmax(0)).all()                                   # the argmax kernel on the kernels' own logits: first max wins
    # forward_labels == argmax(forward) on a fresh stream of the same frames
    e.reset()
    assert e.fifo_len() == 0
    x = weights.synth_video(H, W, 1, seed=1)[0]
    out = np.zeros((1, 19, H, W), np.float32)
    e.forward(x, 0, out)
    e.reset()
    lab = np.zeros((H, W), np.int32)
    e.forward_labels(x, 0, lab)
    assert (lab == out[0].argmax(0)).all()
    e.close()


@pytest.mark.parametrize("name,bb,opts", [("td4", "resnet18", {"attention": 1, "fusion": 63}), ("td2", "resnet18", {"attention": 2, "fusion": 31 + 64}),
                                          ("td2", "resnet18", {"fusion": 6 + 32, "winograd": 4}), ("td2", "resnet18", {"fusion": 63, "winograd": 0})])
def test_pipeline_with_fusion_options_against_reference_goldens(lib, golden_dir, name, bb, opts):
    """tdnet_opts.attention = 1 (online softmax) and every tdnet_opts.fusion bit (q/k projections on the side stream, LayerNorm
    statistics from the attention epilogue, LayerNorm applied inside the head's Winograd input transform, split pyramid row sums)
    against the goldens of the real reference, stage by stage -- including `ln`, which the fused path materialises only on request --
    and with the head on F(4x4) (default scope and every stride-1 3x3) and direct (where bit 4 must fall back to the separate normalisation kernel)."""
    H, W = 33, 65
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    hk, wk = arch.key_size(h), arch.key_size(w)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    e = Engine(spec.path_num, int(bb[6:]), 19, H, W, 0, lib=lib, opts=opts)
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    shapes = {"z": (1, spec.d_model, h, w), "v_cur": (1, spec.d_v, h, w), "q_cur": (1, h * w, 64), "ln": (1, spec.d_v, h, w),
              "lowres": (1, 19, h, w), "cache_q": (1, hk * wk, 64), "cache_k": (1, hk * wk, 64), "cache_v": (1, hk * wk, spec.d_v)}
    T = 2 if "winograd" in opts else spec.path_num + 1                # the fall-back variants: one warm-up + one steady-state frame
    for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
        out = np.full((1, 19, H, W), 7e7, np.float32)
        e.forward(x, t % spec.path_num, out)
        for st, shp in shapes.items():
            key = "f%d_%s" % (t, st)
            if key in g.files:
                got = e.stage(st, shp)
                assert np.abs(got - g[key]).max() <= 1e-4 * max(1.0, np.abs(g[key]).max()), (t, st)
        assert np.abs(out - g["f%d_logits" % t]).max() <= 1e-3
        assert (out[0].argmax(0) == g["f%d_logits" % t][0].argmax(0)).all()
    e.close()


def test_fp16_storage_conv_attention_and_pipeline(lib, golden_dir):
    """tdnet_opts.precision = 1 (BASELINE config 5, "fp16 MFMA"): fp16 activation maps between the backbone's convs (every tile
    variant, 1x1 / 3x3, stride, dilation, ragged shapes, residual), the fp16-MFMA attention kernel (both shapes, ragged, a dominating
    key, the LayerNorm statistics of its epilogue), then the td2 pipeline in that mode against the goldens of the real (fp32)
    reference with the gate this mode is held to: max|dlogit| <= 3e-2 and >= 99.5 % of the labels equal."""
    for H, W in ((33, 65), (18, 23), (7, 9)):           # the 7x7 stem on the fp16 MFMA (two taps per LDS slot, borders on all sides)
        opcheck.stem(lib, MEM, H, W, tol=1e-2, opts={"precision": 1})
    for tile in (3, 4, 5):
        opcheck.conv_f16io(lib, MEM, 13, 21, 128, 96, 3, 1, 1, 1, True, tile)
        opcheck.conv_f16io(lib, MEM, 7, 9, 64, 64, 1, 1, 1, 0, False, tile)
        opcheck.conv_f16io(lib, MEM, 9, 11, 192, 130, 1, 2, 1, 2, True, tile)
        opcheck.conv_f16io(lib, MEM, 12, 17, 64, 128, 3, 2, 1, 1, False, tile)
        opcheck.conv_f16io(lib, MEM, 10, 14, 128, 64, 3, 1, 4, 1, True, tile)
    for ln in (False, True):
        opcheck.attention(lib, MEM, 45, 6, 512, online=16, tol=1e-2, ln=ln)
        opcheck.attention(lib, MEM, 300, 200, 512, spike=True, online=16, tol=1e-2, ln=ln)
        opcheck.attention(lib, MEM, 130, 193, 128, True, True, spike=True, online=16, tol=1e-2, ln=ln)
        opcheck.attention(lib, MEM, 97, 300, 512, ramp=True, online=16, tol=1e-2, ln=ln)
        opcheck.attention(lib, MEM, 33, 1, 128, online=16, tol=1e-2, ln=ln)
    name, bb, H, W = "td2", "resnet18", 33, 65
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    e = Engine(2, 18, 19, H, W, 0, lib=lib, opts={"precision": 1})
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    for t, x in enumerate(weights.synth_video(H, W, 3, seed=1)):
        out = np.full((1, 19, H, W), 7e7, np.float32)
        e.forward(x, t % 2, out)
        ref = g["f%d_logits" % t]
        assert np.abs(out - ref).max() <= 3e-2, (t, np.abs(out - ref).max())
        assert (out[0].argmax(0) == ref[0].argmax(0)).mean() >= 0.995
        assert np.abs(e.stage("c4", (1, 512, h, w)) - g["f%d_c4" % t]).max() <= 3e-2 * np.abs(g["f%d_c4" % t]).max()
    e.close()
    # td4: the head's 3x3 conv has 128 output channels -- LayerNorm writes its map as fp16 and the conv runs on the LDS-DMA kernel; the
    # fp32 "ln" stage is materialised on request from the same statistics
    name = "td4"
    spec = arch.model_spec(name, 19, bb)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    e = Engine(4, 18, 19, H, W, 0, lib=lib, opts={"precision": 1})
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    for t, x in enumerate(weights.synth_video(H, W, 5, seed=1)):
        out = np.full((1, 19, H, W), 7e7, np.float32)
        e.forward(x, t % 4, out)
        ref = g["f%d_logits" % t]
        assert np.abs(out - ref).max() <= 3e-2, (t, np.abs(out - ref).max())
        assert (out[0].argmax(0) == ref[0].argmax(0)).mean() >= 0.995
        ln = e.stage("ln", (1, 512, h, w))
        assert np.abs(ln - g["f%d_ln" % t]).max() <= 3e-2 * np.abs(g["f%d_ln" % t]).max(), t
    e.close()


def test_fp16_conv_on_lds_dma(lib):
    """td_conv_hd.h k_conv_dma_h: the fp16-map convs fed by LDS-DMA -- 256 / 192 / 128-row tiles (tile codes 18 / 17 / 16; 4-, 3- and
    2-buffer rings), the XOR-swizzled activation image and the piece -> wave assignment (RH = 3 repeats two pieces), 3x3 with dilation /
    stride / padding taps on every side, 1x1 with stride, ragged M (rows past the map read zeros) and Cout not a multiple of 128,
    1 .. 27 K steps, residual and activations; then the heuristic's own choice."""
    opcheck.conv_f16io(lib, MEM, 13, 21, 128, 256, 3, 1, 1, 1, True, 19)             # the 256 x 256 tile: a wave multiplies two 64-slot weight groups
    opcheck.conv_f16io(lib, MEM, 20, 23, 64, 512, 3, 1, 2, 2, True, 19)              # two M tiles (ragged), two N tiles
    opcheck.conv_f16io(lib, MEM, 9, 11, 192, 256, 1, 2, 1, 0, False, 19)
    for tile in (18, 17, 16, 20, 21, 22, None):                                       # 20 / 21: 128 rows on a ring of four / two buffers; 22: eight waves of 32 x 64
        opcheck.conv_f16io(lib, MEM, 13, 21, 128, 160, 3, 1, 1, 1, True, tile)       # ragged M and N, two N tiles
        opcheck.conv_f16io(lib, MEM, 7, 9, 64, 128, 1, 1, 1, 0, False, tile)         # a single K step
        opcheck.conv_f16io(lib, MEM, 9, 11, 192, 130, 1, 2, 1, 2, True, tile)        # 1x1 stride 2, three steps, 130 channels
        opcheck.conv_f16io(lib, MEM, 12, 17, 64, 128, 3, 2, 1, 1, False, tile)       # stride-2 3x3
        opcheck.conv_f16io(lib, MEM, 10, 14, 128, 256, 3, 1, 4, 1, True, tile)       # dilation 4: most taps padded
        opcheck.conv_f16io(lib, MEM, 20, 23, 64, 128, 3, 1, 2, 0, False, tile)       # 460 pixels: several M tiles, ragged last one
    # k_conv_dma_h3 (one LDS image per kernel row, the row's three taps read it at shifted slots): every 3x3 stride-1 case above ran on
    # it; tile + 32 keeps the tap-by-tap kernel, and the two must agree BIT FOR BIT (same products, same summation order).  Tiles that
    # span one, two and three image rows, halos of 1 .. 16 columns, a map narrower than the halo, rows past the map, a dilation whose
    # halo does not fit the image buffer (falls back to the tap-by-tap kernel by itself)
    for tile in (16, 17, 18, 19, 20, 22, 23, 24, 25, 26, 27, 28, 29):                 # 23 / 24: one barrier per super-step (128 / 192 rows); 25 / 26: per K step; 27-29: early landing
        for H, W, Cin, Cout, dil in [(13, 21, 128, 256, 1), (10, 14, 128, 256, 4), (5, 300, 64, 256, 2), (3, 130, 64, 256, 8),
                                     (40, 7, 64, 256, 3), (9, 40, 192, 256, 16), (2, 2, 64, 256, 1)]:
            _, a = opcheck.conv_f16io(lib, MEM, H, W, Cin, Cout, 3, 1, dil, 1, True, tile, want_out=True)
            _, b = opcheck.conv_f16io(lib, MEM, H, W, Cin, Cout, 3, 1, dil, 1, True, tile + 32, want_out=True)
            assert np.array_equal(a, b), (tile, H, W, Cin, Cout, dil, np.abs(a - b).max())


def test_fp16_conv_with_dedicated_loader_waves(lib):
    """td_conv_hd.h k_conv_dma_h3p (round 4): the row-image conv with four LOADER waves per workgroup that only issue LDS-DMA, wait and
    meet the barrier, while the matrix waves only read fragments and multiply.  Tile codes 31 / 32 / 33 = 128 (eight matrix waves) / 192 /
    256 rows; 34 / 35 / 36 = the NARROW tiles (rows x 64 channels, k_conv_dma_h3n: the default for small maps).  Same images, same weight
    ring, same products in the same order as the tap-by-tap kernel (tile code 48 + ..): BIT FOR BIT, over tiles that span one to three
    image rows, halos of 1 .. 16 columns (each picks the smallest image buffer that holds it, or falls back to the plain kernel), ragged
    M, rows past the map; then residual / activation variants and the whole model with tdnet_opts.fusion bits 8192 / 32768 against the
    plain fp16 pipeline."""
    for tile, plain in ((31, 54), (32, 49), (33, 50), (34, 54), (35, 49), (36, 50)):
        for H, W, Cin, Cout, dil in [(13, 21, 128, 256, 1), (10, 14, 128, 256, 4), (5, 300, 64, 256, 2), (3, 130, 64, 256, 8),
                                     (40, 7, 64, 256, 3), (9, 40, 192, 256, 16), (2, 2, 64, 256, 1), (23, 37, 64, 160, 2)]:
            _, a = opcheck.conv_f16io(lib, MEM, H, W, Cin, Cout, 3, 1, dil, 1, True, tile, want_out=True)
            _, b = opcheck.conv_f16io(lib, MEM, H, W, Cin, Cout, 3, 1, dil, 1, True, plain, want_out=True)
            assert np.array_equal(a, b), (tile, H, W, Cin, Cout, dil, np.abs(a - b).max())
        opcheck.conv_f16io(lib, MEM, 13, 21, 128, 130, 3, 1, 1, 0, False, tile)        # no residual, no activation, ragged N
        opcheck.conv_f16io(lib, MEM, 9, 11, 192, 256, 1, 1, 1, 2, True, tile)          # 1x1: not a row-image conv, the plain kernel by itself
        opcheck.conv_f16io(lib, MEM, 12, 17, 64, 128, 3, 2, 1, 1, False, tile)         # stride 2 likewise
    H, W = 33, 65
    spec = arch.model_spec("td2", 19, "resnet34")
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    outs, launches = [], []
    for fusion in (38, 38 | 8192, 38 | 8192 | 32768, 38 | 131072):     # 131072: the Encoding's five 1x1 convs as two grouped launches (round 5)
        e = Engine(2, 34, 19, H, W, 0, lib=lib, opts={"precision": 1, "fusion": fusion})
        e.load_state_dict(sd)
        o = []
        for t, x in enumerate(weights.synth_video(H, W, 3, seed=1)):
            out = np.zeros((1, 19, H, W), np.float32)
            e.forward(x, t % 2, out)
            o.append(out)
        outs.append(o)
        launches.append(e.last_launch_count())
        e.close()
    assert all(np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d) for a, b, c, d in zip(*outs))
    # the Encoding: 5 launches -> 2; conv1 + downsample of layers 2.0 / 3.0 / 4.0 in one launch each (on this tiny map layers 3-4 stay on the
    # register-staged kernel too; at 720x960 their conv1 is an LDS-DMA conv and only layer2.0 pairs up)
    assert launches[3] == launches[0] - 3 - 3, launches


@pytest.mark.parametrize("name,P,layers,bb", [("td4", 4, 18, "resnet18"), ("td2", 2, 50, "resnet50")])
def test_fp16_grouped_launches_equal_single_launches(lib, name, P, layers, bb):
    """tdnet_opts.fusion bit 131072 (fp16 default since round 5, td_conv_h.h k_conv_igemm_h_group): up to three independent convs of one
    kernel form in ONE grid -- the Encoding's value / query / key first layers, then the query / key second layers; a BasicBlock's conv1
    beside its 1x1 downsample.  Same body, same products, same order: logits bit for bit through warm-up and steady state, with the
    launch count down by 3 (Encoding) + 2 (a BasicBlock backbone's downsample pairs whose conv1 runs on the register-staged kernel: on this
    small map layer2.0 and layer4.0 -- layer3.0's conv1 is a 256-channel "same" conv on the narrow LDS-DMA tiles, fusion bit 32768; a
    Bottleneck backbone has no pair: its conv1 is not the downsample's sibling)."""
    H, W = 33, 65
    spec = arch.model_spec(name, 19, bb)
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    default = lib.opts(precision=1).fusion
    assert default & 131072
    outs, launches = [], []
    for fusion in (default, default & ~131072):
        e = Engine(P, layers, 19, H, W, 0, lib=lib, opts={"precision": 1, "fusion": fusion})
        e.load_state_dict(sd)
        o = []
        for t, x in enumerate(weights.synth_video(H, W, P + 2, seed=4)):
            out = np.zeros((1, 19, H, W), np.float32)
            e.forward(x, t % P, out)
            o.append(out)
        outs.append(o)
        launches.append(e.last_launch_count())
        e.close()
    assert all(np.array_equal(a, b) for a, b in zip(*outs))
    assert launches[1] - launches[0] == (5 if layers < 50 else 3), launches


def test_winograd_f4_conv_and_pipeline(lib, golden_dir):
    """Winograd F(4x4,3x3) (td_wino.h k_wino4_in / k_wino4_out, 36 batched GEMMs): every dilation, ragged sizes (tiles hanging
    over the image, images smaller than a tile), residual/activation variants, then the td4 pipeline with layers 3-4 and the
    head on it (mode 3, the wide convs) against the goldens captured from the real reference."""
    if True:
        worst = 0.0
        for a in [(13, 21, 64, 128, 3, 1, 2, 1, True), (12, 30, 64, 160, 3, 1, 4, 1, False), (9, 17, 128, 256, 3, 1, 8, 0, True),
                  (5, 9, 256, 512, 3, 1, 16, 2, False), (40, 40, 32, 128, 3, 1, 1, 1, False), (1, 1, 32, 32, 3, 1, 1, 0, False),
                  (7, 7, 32, 64, 3, 1, 3, 1, True), (16, 32, 64, 64, 3, 1, 1, 2, True)]:
            worst = max(worst, opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4}))       # F4's per-conv error is ~6x F2's; outputs are O(1)
            opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4, "fusion": 64})             # padded workspace planes
        opcheck.conv(lib, MEM, 13, 21, 32, 96, 3, 2, 1, 0, False, opts={"winograd": 4})           # stride 2 is not eligible: direct path
        name, bb, H, W = "td4", "resnet18", 33, 65
        spec = arch.model_spec(name, 19, bb)
        h, w = arch.feat_size(H), arch.feat_size(W)
        g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
        e = Engine(4, 18, 19, H, W, 0, lib=lib, opts={"winograd": 3})
        e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
        for t, x in enumerate(weights.synth_video(H, W, 4, seed=1)):
            out = np.full((1, 19, H, W), 7e7, np.float32)
            e.forward(x, t % 4, out)
            assert np.abs(e.stage("c4", (1, 512, h, w)) - g["f%d_c4" % t]).max() <= 1e-4 * np.abs(g["f%d_c4" % t]).max()
            assert np.abs(out - g["f%d_logits" % t]).max() <= 1e-3
            assert (out[0].argmax(0) == g["f%d_logits" % t][0].argmax(0)).all()
        e.close()


def test_winograd_chunked_low_register_transforms(lib):
    """td_wino.h k_wino4_in_c / k_wino4_out_c (tdnet_opts.overlap): the wave-per-(tile, channel slice) F(4x4) transforms with 1, 2 and 4
    channels per lane, whole convs (bit 2) and as the two row-parity chunks of an even-dilation conv (bit 1, the chunks run one after the
    other here): every dilation, images smaller than a tile, odd heights (the two parities have different row counts), channel counts
    that are not a multiple of the 64-lane slice, residual / activation variants."""
    shapes = [(13, 21, 64, 128, 3, 1, 2, 1, True), (12, 30, 64, 160, 3, 1, 4, 1, False), (9, 17, 128, 256, 3, 1, 8, 0, True),
              (5, 9, 256, 512, 3, 1, 16, 2, False), (40, 40, 32, 128, 3, 1, 1, 1, False), (1, 1, 32, 32, 3, 1, 1, 0, False),
              (7, 7, 32, 64, 3, 1, 3, 1, True), (16, 32, 64, 64, 3, 1, 1, 2, True), (3, 5, 96, 36, 3, 1, 2, 1, True)]
    for vw in (0, 1, 2):
        for a in shapes:
            opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4, "overlap": 2 | (vw << 4)})       # whole conv on the new kernels
            opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4, "overlap": 1 | (vw << 4)})       # even dilation: two chunks
            if a[6] % 4 == 0:
                opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4, "overlap": 1 | 64 | (vw << 4)})  # dilation 4 / 8 / 16: four row classes mod 4 (round 5's Infinity-Cache probe hook)
    opcheck.conv(lib, MEM, 12, 30, 64, 160, 3, 1, 4, 1, True, tol=2e-4, opts={"winograd": 4, "overlap": 1, "gemm_persistent": 3, "fusion": 64})
    # td_gemm_dma.h (overlap bit 8): the batched GEMMs fed by LDS-DMA -- K = 32 .. 256 (1 .. 8 steps), ragged M and N, several tiles per
    # workgroup (grid forced small), padded planes, whole convs and chunks; bit-identical to the register-staged GEMM
    for a in shapes:
        e0 = opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4, "overlap": 1})
        e1 = opcheck.conv(lib, MEM, *a, tol=2e-4, opts={"winograd": 4, "overlap": 1 | 8})
        assert e0 == e1, (a, e0, e1)
    for cap in (2, 3, 5, 8):
        opcheck.conv(lib, MEM, 12, 30, 64, 160, 3, 1, 4, 1, True, tol=2e-4, opts={"winograd": 4, "overlap": 8, "gemm_persistent": cap, "fusion": 64})
        opcheck.conv(lib, MEM, 13, 21, 96, 128, 3, 1, 2, 1, True, tol=2e-4, opts={"winograd": 4, "overlap": 9 | 16, "gemm_persistent": cap})


@pytest.mark.parametrize("name,bb,opts", [("td4", "resnet18", {"overlap": 41 | 4}), ("td4", "resnet18", {"overlap": 0}), ("td4", "resnet34", {"overlap": 1 | 4 | 16}),
                                          ("td2", "resnet18", {"overlap": 3 | 4 | 32}), ("td2", "resnet18", {"overlap": 1 | 4 | 8, "gemm_persistent": 5}),
                                          ("td2", "resnet18", {"overlap": 41})])
def test_pipeline_row_parity_chains(lib, golden_dir, name, bb, opts):
    """tdnet_opts.overlap: layers 3-4 as an even-row and an odd-row chain of Winograd convs (+ the 1x1 downsample on image rows), 1 / 2 /
    4 channels per lane in the transforms, the LDS-DMA-fed GEMM (bit 8; 41 = the library default, also with several tiles per workgroup)
    against the reference goldens; `c4` is read from the run's own block buffers.  The feature map is 5 x 9 here: 3 even rows, 2 odd.
    (The schedules that lost in rounds 3-4 -- staggered start, riders, pre-launched chain, CU-mask partition -- were removed in round 5.)"""
    H, W = 33, 65
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    e = Engine(spec.path_num, int(bb[6:]), 19, H, W, 0, lib=lib, opts=opts)
    assert e.opts()["overlap"] == opts["overlap"]
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    # bit 4 forces the chains on this 5 x 9 map (by default they start at 24000 feature pixels: the last case runs unchained); a chained frame
    # issues every conv of the run as two launches
    for t, x in enumerate(weights.synth_video(H, W, spec.path_num + 1, seed=1)):
        out = np.full((1, 19, H, W), 7e7, np.float32)
        e.forward(x, t % spec.path_num, out)
        if "f%d_c4" % t in g.files:
            assert np.abs(e.stage("c4", (1, 512, h, w)) - g["f%d_c4" % t]).max() <= 1e-4 * np.abs(g["f%d_c4" % t]).max()
        assert np.abs(out - g["f%d_logits" % t]).max() <= 1e-3
        assert (out[0].argmax(0) == g["f%d_logits" % t][0].argmax(0)).all()
        if t == 0:
            n_chained = e.last_launch_count()
    e.close()
    if opts["overlap"] & 1:                                                      # the chains really ran (or, without bit 4 on a map this small, really did not)
        e0 = Engine(spec.path_num, int(bb[6:]), 19, H, W, 0, lib=lib, opts=dict(opts, overlap=opts["overlap"] & ~5))
        e0.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
        e0.forward(weights.synth_video(H, W, 1, seed=1)[0], 0, np.zeros((1, 19, H, W), np.float32))      # frame 0 of the same clip
        assert (n_chained > e0.last_launch_count()) == bool(opts["overlap"] & 4), (n_chained, e0.last_launch_count())
        e0.close()


def test_activation_propagates_non_finite_values_like_the_reference(lib):
    """td_conv.h td_activate (shared by the conv / GEMM epilogues and the Winograd output transforms): a NaN in the input reaches the
    output as NaN under every activation (F.relu / F.leaky_relu do the same; a max/min formulation would turn it into 0), ReLU(-inf)
    is 0, LeakyReLU(-inf) and no-activation(-inf) stay -inf.  Direct conv, persistent 1x1 GEMM and F(4x4) Winograd."""
    import ctypes
    H, W, Cin, Cout = 6, 10, 64, 128
    g = np.random.default_rng(3)
    x = g.standard_normal((H, W, Cin)).astype(np.float32)
    x[2, 3, 5] = np.nan
    for KS, o in ((1, {}), (3, {"winograd": 0}), (3, {"winograd": 4, "overlap": 2})):
        w = (g.standard_normal((Cout, Cin, KS, KS)) / np.sqrt(Cin * KS * KS)).astype(np.float32)
        b = g.standard_normal(Cout).astype(np.float32)
        b[7] = -np.inf
        for act in (0, 1, 2):
            out = MEM.empty((H, W, Cout))
            lib.check(lib.tdnet_op_conv2d(MEM.ptr(MEM.put(x)), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, KS, 1, 1, None, act,
                                          ctypes.byref(lib.opts(**o)), -1, MEM.ptr(out), None))
            reach = KS // 2 if not o.get("winograd") else 5                     # a Winograd tile spreads the NaN over its 4x4 outputs
            assert np.isnan(out[2, 3]).all(), (KS, o, act)
            clean = np.ones((H, W), bool)
            clean[max(0, 2 - reach):2 + reach + 1, max(0, 3 - reach):3 + reach + 1] = False
            assert np.isfinite(np.delete(out[clean], 7, axis=1)).all(), (KS, o, act)
            col = out[clean][:, 7]
            assert (col == 0).all() if act == 1 else np.isneginf(col).all(), (KS, o, act, col[:4])


def test_persistent_gemm_multi_tile(lib):
    """td_gemm.h: stride-1 1x1 convs and Winograd GEMMs on the persistent kernel, with the grid forced small so every
    workgroup walks several tiles (pipeline running across tile boundaries, odd/even tile counts, idle workgroups)."""
    for cap in (1, 3, 5, 8, 11):
        pers = cap if cap > 1 else 2
        for tile in (3, 4, 5):
            o = {"gemm_persistent": pers}
            opcheck.conv(lib, MEM, 23, 31, 128, 160, 1, 1, 1, 1, True, tile, opts=o)     # 4 K steps of 32, ragged M and N
            opcheck.conv(lib, MEM, 40, 40, 64, 64, 1, 1, 1, 0, False, tile, opts=o)      # one period per tile
            opcheck.conv(lib, MEM, 23, 31, 96, 160, 1, 1, 1, 1, True, tile, opts=o)      # K = 96: odd step count -> single-tile kernel
        opcheck.conv(lib, MEM, 12, 30, 64, 160, 3, 1, 4, 1, True, tol=2e-4, opts={"gemm_persistent": pers, "winograd": 4})   # 36 batches
    opcheck.conv(lib, MEM, 23, 31, 128, 160, 1, 1, 1, 1, True, opts={"gemm_persistent": 0})   # the non-persistent fallback stays correct
    opcheck.conv(lib, MEM, 12, 30, 64, 160, 3, 1, 4, 1, True, tol=2e-4, opts={"gemm_persistent": 0, "winograd": 4})


def test_two_handles_with_different_options_coexist(lib, golden_dir):
    """Nothing is process-wide: an all-direct handle and a Winograd F(4x4) handle created side by side keep their own configuration
    and both meet the gate on the reference goldens (frames interleaved between the two handles)."""
    name, bb, H, W = "td2", "resnet18", 33, 65
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    ea = Engine(2, 18, 19, H, W, 0, lib=lib, opts={"winograd": 0})
    eb = Engine(2, 18, 19, H, W, 0, lib=lib, opts={"winograd": 4, "pipeline": 0})
    assert ea.opts()["winograd"] == 0 and eb.opts()["winograd"] == 4 and eb.opts()["pipeline"] == 0 and ea.opts()["pipeline"] == 1
    sd = weights.synth_state_dict(spec, h, w, 0)
    ea.load_state_dict(sd); eb.load_state_dict(sd)
    for t, x in enumerate(weights.synth_video(H, W, 2, seed=1)):
        for e in (ea, eb):
            out = np.full((1, 19, H, W), 7e7, np.float32)
            e.forward(x, t % 2, out)
            assert np.abs(out - g["f%d_logits" % t]).max() <= 1e-3
    ea.close(); eb.close()


def test_shared_weight_block_handles(lib, golden_dir):
    """include/tdnet.h tdnet_create_shared: a second handle on the SAME weight block (own workspace, FIFO, streams) is another video
    stream -- bit for bit what a handle with its own copy of the weights computes, interleaved with the owner's stream -- costs no
    second copy of the weights (tdnet_memory_bytes), refuses other options than the block was packed for, and outlives the handle that
    loaded the weights (reference count: the OWNER is destroyed first here).  td4_psp18.py:216-229: N samples, one module."""
    import ctypes
    name, bb, H, W = "td4", "resnet18", 33, 65
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = weights.synth_state_dict(spec, h, w, 0)
    owner = Engine(4, 18, 19, H, W, 0, lib=lib)
    with pytest.raises(_capi.TdnetError):
        owner.share()                                                         # not finalized yet
    owner.load_state_dict(sd)
    alone = Engine(4, 18, 19, H, W, 0, lib=lib)
    alone.load_state_dict(sd)
    shared = owner.share()
    wb, hb, refs = owner.memory_bytes()
    wb2, hb2, refs2 = shared.memory_bytes()
    assert refs == refs2 == 2 and wb == wb2 > 0 and hb == hb2 > 0             # one block, two equal workspaces
    assert alone.memory_bytes() == (wb, hb, 1)
    assert shared.opts() == owner.opts()
    hh = ctypes.c_void_p()
    o = lib.opts(winograd=0)
    assert lib.tdnet_create_shared(owner.h, ctypes.byref(o), ctypes.byref(hh)) < 0 and b"tdnet_opts differ" in lib.tdnet_last_error()
    assert lib.tdnet_create_shared(owner.h, ctypes.byref(lib.opts()), ctypes.byref(hh)) == 0          # the same options spelled out: accepted
    lib.tdnet_destroy(hh)
    assert owner.memory_bytes()[2] == 2
    va, vb = weights.synth_video(H, W, 6, seed=1), weights.synth_video(H, W, 6, seed=2)
    outs = {}
    for t in range(6):                                                        # the two streams interleaved; the owner dies after frame 2
        for tag, e, x in (("a", owner, va[t]), ("b", shared, vb[t]), ("ref", alone, vb[t])):
            if e is None:
                continue
            out = np.full((1, 19, H, W), 7e7, np.float32)
            e.forward(x, t % 4, out)
            outs[(tag, t)] = out
        assert np.array_equal(outs[("b", t)], outs[("ref", t)]), t
        if t == 2:
            owner.close()
            owner = None
            assert shared.memory_bytes() == (wb, hb, 1)
    assert shared.last_launch_count() > 20 and shared.last_launch_count() == alone.last_launch_count()
    lib.check(lib.tdnet_warmup(shared.h, None))                               # idempotent, nothing to place in the emulator
    shared.close(); alone.close()


@pytest.mark.parametrize("name,T", [("td4", 4), ("td2", 2)])
def test_split_frame_and_cache_transport_equal_the_single_handle_stream(lib, name, T):
    """include/tdnet.h split API: two handles play two path-parallel ranks (frames t = g mod 2), exchanging cache entries
    with tdnet_cache_export / tdnet_cache_push in the order of parallel.PathParallelStream.  Outputs and FIFO states must be
    BIT-identical to one handle running tdnet_forward over the stream."""
    H, W, bb = 33, 65, "resnet18"
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = weights.synth_state_dict(spec, h, w, 0)
    frames = weights.synth_video(H, W, T, seed=3)
    one = Engine(spec.path_num, 18, 19, H, W, 0, lib=lib)
    one.load_state_dict(sd)
    ref = []
    for t, x in enumerate(frames):
        out = np.zeros((1, 19, H, W), np.float32)
        one.forward(x, t % spec.path_num, out)
        ref.append(out)
    ranks = [Engine(spec.path_num, 18, 19, H, W, 0, lib=lib) for _ in range(2)]
    for e in ranks:
        e.load_state_dict(sd)
    lk, dk, dv = ranks[0].cache_dims()
    assert (lk, dk, dv) == (arch.key_size(h) * arch.key_size(w), 64, spec.d_v)
    got = {}
    for r0 in range(0, T, 2):
        n_valid = min(2, T - r0)
        entries = []
        for g in range(n_valid):                                                  # 1. every rank encodes its own frame
            ranks[g].encode(frames[r0 + g], (r0 + g) % spec.path_num)
            q, k, v = np.zeros((lk, dk), np.float32), np.zeros((lk, dk), np.float32), np.zeros((lk, dv), np.float32)
            ranks[g].cache_export(q, k, v)                                        # 2. the exchange
            entries.append((q, k, v))
        for g in range(n_valid):                                                  # 3. walk the round in frame order
            for j in range(n_valid):
                if j == g:
                    out = np.zeros((1, 19, H, W), np.float32)
                    ranks[g].propagate(out)
                    got[r0 + g] = out
                else:
                    ranks[g].cache_push(*entries[j])
    for t in range(T):
        assert np.array_equal(got[t], ref[t]), t
    # the split halves on ONE handle are tdnet_forward; a second encode before propagate is refused, as is a stray propagate
    one.reset()
    out = np.zeros((1, 19, H, W), np.float32)
    one.encode(frames[0], 0)
    with pytest.raises(Exception):
        one.encode(frames[1], 1 % spec.path_num)
    one.propagate(out)
    with pytest.raises(Exception):
        one.propagate(out)
    assert np.array_equal(out, ref[0])
    for e in ranks + [one]:
        e.close()


def test_split_gemm_precision2_ops(lib):
    """tdnet_opts.precision = 2 / 3 (td_gemm_b3.h): fp32 operands as three bf16 parts, six bf16-MFMA products, fp32 accumulate.  Operator
    level on the emulator: the Winograd GEMMs (whole conv and row-parity chunks, dilations, ragged tile counts, N not a multiple of 128, N = 64
    with half a tile of padding), the stride-1 1x1 conv with bias / residual / activations (all three epilogue roles), K = 32 .. 192 (2 .. 12
    steps of 16: the two-buffer pipeline's prologue and both parities), several tiles per workgroup.  Tolerance of the fp32 kernels."""
    o3 = {"precision": 3}
    opcheck.conv(lib, MEM, 13, 21, 128, 128, 3, 1, 2, 1, True, opts=o3)                # dilated 3x3 + residual + ReLU (Winograd, 36 GEMMs)
    opcheck.conv(lib, MEM, 9, 17, 128, 256, 3, 1, 1, 0, False, opts=o3)                # two N tiles
    opcheck.conv(lib, MEM, 20, 30, 256, 132, 3, 1, 4, 1, True, opts=dict(o3, overlap=41 | 4))   # row-parity chunks, ragged N
    opcheck.conv(lib, MEM, 11, 19, 64, 160, 1, 1, 1, 2, True, opts=o3)                 # 1x1: residual + LeakyReLU (ROLE 0)
    opcheck.conv(lib, MEM, 40, 40, 128, 64, 1, 1, 1, 1, False, opts=o3)                # 1x1 to 64 channels (padded to one 128-column tile), ReLU (ROLE 2)
    opcheck.conv(lib, MEM, 33, 9, 192, 128, 1, 1, 1, 0, False, opts=o3)                # K = 192: twelve steps; 297 rows: two M tiles, the second ragged
    opcheck.conv(lib, MEM, 7, 9, 64, 128, 1, 1, 1, 0, True, opts=o3)                   # K = 64: four steps, one ragged tile
    opcheck.conv(lib, MEM, 70, 70, 64, 256, 1, 1, 1, 0, False, opts=dict(o3, gemm_persistent=3))   # 40 tiles on three workgroups
    opcheck.conv(lib, MEM, 24, 24, 128, 128, 3, 1, 1, 1, False, opts=dict(o3, gemm_persistent=5))  # 36 batches walked by five workgroups
    # the attention kernel of precision 2 (td_attn_b3.h, online = 17): q, k, P and v' as three bf16 parts; Lk below one key tile, ragged super-tiles
    # with a dominating key, the d_v = 128 variant, LayerNorm strip statistics from the epilogue, keys sorted by growing score
    opcheck.attention(lib, MEM, 45, 6, 512, online=17)
    opcheck.attention(lib, MEM, 300, 200, 512, spike=True, online=17)
    opcheck.attention(lib, MEM, 200, 131, 128, True, False, qk_scale=2.0, online=17)
    opcheck.attention(lib, MEM, 130, 193, 128, True, True, spike=True, online=17, ln=True)
    opcheck.attention(lib, MEM, 70, 300, 512, online=17, ramp=True, ln=True)
    # the 64-query / eight-wave form of the 512-channel kernel (k_attention_b3w; picked by a frame for Lq >= 16384, forced here by online = 18): a workgroup's second
    # query tile empty, partly filled and full; strips past Lq not written; a single key; keys sorted by growing score (the reference moves in both query tiles)
    opcheck.attention(lib, MEM, 45, 6, 512, online=18)
    opcheck.attention(lib, MEM, 300, 200, 512, spike=True, online=18)
    opcheck.attention(lib, MEM, 97, 130, 512, True, True, spike=True, online=18, ln=True)
    opcheck.attention(lib, MEM, 70, 300, 512, online=18, ramp=True, ln=True)
    opcheck.attention(lib, MEM, 64, 128, 512, online=18, ln=True)
    opcheck.attention(lib, MEM, 33, 1, 512, online=18, ln=True)
    # the Cout <= 64 convs of precision 2 (td_conv_ad_b3.h, fusion bit 524288: A straight from global memory, weights by LDS-DMA on three buffers): ResNet
    # layer1's shape, an odd number of K steps with stride / dilation / ragged channels, a single step, the strided 1x1 form, the packed-row 7x7 stem
    o2 = {"precision": 2}
    opcheck.conv(lib, MEM, 13, 21, 64, 64, 3, 1, 1, 1, True, opts=o2)
    opcheck.conv(lib, MEM, 11, 9, 96, 48, 3, 2, 2, 0, False, opts=o2)
    opcheck.conv(lib, MEM, 9, 17, 32, 64, 1, 1, 1, 2, True, opts=dict(o2, gemm_persistent=0))
    opcheck.conv(lib, MEM, 9, 17, 64, 40, 1, 2, 1, 2, True, opts=o2)
    for H, W in ((33, 65), (18, 23), (7, 9)):
        opcheck.stem(lib, MEM, H, W, opts=o2)
    opcheck.conv(lib, MEM, 13, 21, 64, 128, 3, 2, 1, 1, False, opts=dict(o2, winograd=0))   # 65 .. 128 output channels: two 64-column tiles (layer2.0's strided conv)
    opcheck.conv(lib, MEM, 9, 17, 64, 100, 1, 2, 1, 0, True, opts=o2)                       # the strided 1x1 downsample, ragged second tile
    opcheck.conv(lib, MEM, 96, 96, 64, 64, 1, 1, 1, 1, True, opts=o2)                       # a stride-1 1x1 conv to <= 128 channels on >= 8192 pixels: kept off the GEMM route
    # the size heuristic of precision 2: a GEMM of fewer than 256 tiles stays on the exact-fp32 kernels, bit for bit
    import ctypes
    g = np.random.default_rng(3)
    x = g.standard_normal((13, 21, 128)).astype(np.float32)
    w = (g.standard_normal((128, 128, 3, 3)) / 34.0).astype(np.float32)
    outs = []
    for prec in (0, 2, 3):
        out = MEM.empty((13, 21, 128))
        oo = lib.opts(precision=prec)
        lib.check(lib.tdnet_op_conv2d(MEM.ptr(MEM.put(x)), 13, 21, 128, w.ctypes.data, None, 128, 3, 1, 1, None, 0, ctypes.byref(oo), -1, MEM.ptr(out), None))
        outs.append(out.copy())
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
    assert np.abs(outs[0] - outs[2]).max() < 1e-4
    x = g.standard_normal((13, 21, 64)).astype(np.float32)
    w = (g.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32)
    outs = []
    for kw in ({}, {"precision": 2}, {"precision": 2, "fusion": lib.opts().fusion & ~524288}):   # the narrow convs: split with the bit, exact fp32 without
        out = MEM.empty((13, 21, 64))
        oo = lib.opts(**kw)
        lib.check(lib.tdnet_op_conv2d(MEM.ptr(MEM.put(x)), 13, 21, 64, w.ctypes.data, None, 64, 3, 1, 1, None, 0, ctypes.byref(oo), -1, MEM.ptr(out), None))
        outs.append(out.copy())
    assert not np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and np.abs(outs[0] - outs[1]).max() < 1e-4


@pytest.mark.parametrize("name,bb,H,W,opts", [("td2", "resnet18", 33, 65, {"precision": 3, "overlap": 41 | 4}), ("td2", "resnet50", 33, 65, {"precision": 3})])
def test_pipeline_precision2_against_reference_goldens(lib, golden_dir, name, bb, H, W, opts):
    """The whole frame with every eligible GEMM on the split kernel (precision 3 = precision 2 at any GEMM size): Winograd convs of layers 2-4 and
    the head -- whole and as row-parity chains, whose downsample conv is the weight-shared batched form --, the Encoding's value conv, the
    attention's fc on the cached value matrix, a Bottleneck backbone's 1x1 convs.  Same gate as the fp32 pipeline against the goldens captured
    from the real reference: every stage within 1e-4 (relative to the stage's scale), logits within 1e-3, labels in the tie band.  (td2: the
    warm-up frame and the first steady-state one -- the kernels are td4's, and a 256 x 128 tile is slow on the emulator whatever it holds.)"""
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    T = spec.path_num
    e = Engine(spec.path_num, int(bb[6:]), 19, H, W, 0, lib=lib, opts=opts)
    assert e.opts()["precision"] == 3
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    shapes = {"c4": (1, spec.d_model, h, w), "z": (1, spec.d_model, h, w), "v_cur": (1, spec.d_v, h, w), "ln": (1, spec.d_v, h, w), "lowres": (1, 19, h, w)}
    for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
        out = np.full((1, 19, H, W), 7e7, np.float32)
        e.forward(x, t % spec.path_num, out)
        for st, shp in shapes.items():
            if "f%d_%s" % (t, st) in g.files:
                ref = g["f%d_%s" % (t, st)]
                assert np.abs(e.stage(st, shp) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (t, st)
        ref = g["f%d_logits" % t]
        err = float(np.abs(out - ref).max())
        assert err <= 1e-3, (t, err)
        bad = out[0].argmax(0) != ref[0].argmax(0)
        if bad.any():
            top2 = np.sort(ref[0], axis=0)[-2:]
            assert ((top2[1] - top2[0])[bad] <= 2 * err).all(), (t, int(bad.sum()))
    e.close()


@pytest.mark.parametrize("name,P,bb,opts", [("td4", 4, "resnet18", {}), ("td2", 2, "resnet18", {"winograd": 4})])
def test_classifier_inside_the_head_output_transform(lib, name, P, bb, opts):
    """tdnet_opts.fusion bit 262144 (round 6, td_wino.h k_wino4_out_cls): the FCN head's 1x1 classifier (td4_psp18.py:295-299) computed by the wave that holds a
    tile's 16 pixels x all hidden channels, with k_classifier's summation order: logits bit for bit those of the two-kernel form, one launch fewer, in the
    warm-up and the steady-state frames.  td4: 128 hidden channels (two per lane).  td2's head (128 -> 64) is a direct conv by default; under winograd = 4
    (F(4x4) for every stride-1 3x3) it exercises the one-channel-per-lane form."""
    H, W = 33, 65
    spec = arch.model_spec(name, 19, bb)
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    base = lib.opts(**opts).fusion
    assert base & 262144                                               # the default since round 6 (profiles/r06p_*)
    outs, launches = [], []
    for fusion in (base & ~262144, base | 262144):
        e = Engine(P, int(bb[6:]), 19, H, W, 0, lib=lib, opts=dict(opts, fusion=fusion))
        e.load_state_dict(sd)
        o = []
        for t, x in enumerate(weights.synth_video(H, W, P, seed=2)):
            out = np.zeros((1, 19, H, W), np.float32)
            e.forward(x, t % P, out)
            o.append(out)
        outs.append(o)
        launches.append(e.last_launch_count())
        e.close()
    assert all(np.array_equal(a, b) for a, b in zip(*outs))
    assert launches[0] - launches[1] == 1, launches


def test_cache_chain_forked_in_front_of_layer3_is_bit_identical(lib):
    """tdnet_opts.fusion bit 1048576 (round 6, default for fp32 / precision 2): the cache-only attention chain of a steady-state frame is enqueued on the side stream in
    front of the backbone's first dilated block instead of at the frame's start (td_frame.h forward_lowres_impl).  Same kernels on the same inputs: warm-up and
    steady-state frames bit for bit, same launch count -- fp32 with the row-parity chains forced on (three streams), and the fp16 mode, which ignores the bit."""
    H, W = 33, 65
    spec = arch.model_spec("td4", 19, "resnet18")
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    for opts in ({"overlap": 41 | 4}, {"precision": 1}):
        base = lib.opts(**opts).fusion
        assert base & 1048576
        outs, launches = [], []
        for fusion in (base & ~1048576, base):
            e = Engine(4, 18, 19, H, W, 0, lib=lib, opts=dict(opts, fusion=fusion))
            e.load_state_dict(sd)
            o = []
            for t, x in enumerate(weights.synth_video(H, W, 6, seed=2)):
                out = np.zeros((1, 19, H, W), np.float32)
                e.forward(x, t % 4, out)
                o.append(out)
            outs.append(o)
            launches.append(e.last_launch_count())
            e.close()
        assert all(np.array_equal(a, b) for a, b in zip(*outs)), opts
        assert launches[0] == launches[1], launches


def test_precision2_with_the_other_options_switched_off(lib):
    """tdnet_opts.precision = 2 does not rely on the defaults around it: all-direct convs (no Winograd GEMMs to split: the narrow convs and the attention remain), no
    persistent GEMMs (the split GEMM needs them: its layers fall back to the exact-fp32 kernels), no fusion bits (no A-from-global kernels, so no split direct conv;
    unfused LayerNorm, separate classifier).  Every combination is a working handle whose logits stay within the fp32 gate of the default's."""
    H, W = 33, 65
    spec = arch.model_spec("td4", 19, "resnet18")
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    ref = None
    for opts in ({}, {"precision": 2, "winograd": 0}, {"precision": 2, "gemm_persistent": 0}, {"precision": 2, "fusion": 0}):
        e = Engine(4, 18, 19, H, W, 0, lib=lib, opts=opts)
        e.load_state_dict(sd)
        outs = []
        for t, x in enumerate(weights.synth_video(H, W, 5, seed=2)):
            out = np.zeros((1, 19, H, W), np.float32)
            e.forward(x, t % 4, out)
            outs.append(out)
        e.close()
        if ref is None:
            ref = outs
        else:
            d = max(float(np.abs(a - b).max()) for a, b in zip(outs, ref))
            assert 0.0 < d < 1e-3, (opts, d)

