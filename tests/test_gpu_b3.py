"""GPU: tdnet_opts.precision = 2 -- the frame's large GEMMs with fp32 operands as three bf16 parts, six bf16-MFMA products, fp32 accumulate
(td_gemm_b3.h).  Opt-in; the mode is held to the UNCHANGED fp32 gates:
  * operator level: error against an fp64 evaluation no larger than the exact-fp32 kernels' on the same inputs (x 1.25 for the draw);
  * the calibrated clips: max|dlogit| <= 1e-3 and label flips only inside the reference's tie band, mIoU >= 0.9995 (test_gpu_model._vs_oracle), at
    1024x2048 (BASELINE configs[2]), 769x1537 (the checkpoint's geometry) and, with the split kernel forced at every GEMM size (precision 3),
    at a mid size that exercises ragged tiles;
  * SURVEY 8d's un-calibrated init: max error <= 4x and rms <= 3x the fp32 CPU path's own distance to an fp64 evaluation, at 129x257 (three clips
    pooled, split kernel forced) and at 769x1537 -- with the exact-fp32 kernels' figures printed beside them (Testing/model/pspnet/resnet.py:25-59,
    transformer.py:126-139 are fp32 throughout; this mode's products are accurate to 2^-26, its sums are not the fp32 MFMA's bit for bit)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import opcheck
from tdnet_amd import _capi
from test_gpu_model import _reference_init_stress, _vs_oracle

pytestmark = pytest.mark.gpu


def _conv_errors(lib, H, W, Cin, Cout, KS, dil, resid, act, variants, seed=0):
    g = np.random.default_rng(seed)
    x = g.standard_normal((H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((Cout, Cin, KS, KS)) / np.sqrt(Cin * KS * KS)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double().permute(2, 0, 1)[None], torch.from_numpy(w).double(), torch.from_numpy(b).double(), 1, dil * (KS // 2), dil)
    r = None
    if resid:
        r = g.standard_normal((H, W, Cout)).astype(np.float32)
        ref = ref + torch.from_numpy(r).double().permute(2, 0, 1)[None]
    if act == 1:
        ref = F.relu(ref)
    ref = ref[0].permute(1, 2, 0).numpy()
    dx = torch.from_numpy(x).cuda()
    dr = torch.from_numpy(r).cuda() if resid else None
    errs = []
    for kw, tile in variants:
        out = torch.full((H, W, Cout), 7e7, device="cuda")
        o = lib.opts(**kw)
        lib.check(lib.tdnet_op_conv2d(dx.data_ptr(), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, KS, 1, dil, None if dr is None else dr.data_ptr(), act,
                                      ctypes.byref(o), tile, out.data_ptr(), None))
        e = np.abs(out.cpu().numpy().astype(np.float64) - ref)
        errs.append((float(e.max()), float(np.sqrt((e ** 2).mean()))))
    return errs


def test_split_gemm_operators_are_as_accurate_as_the_fp32_kernels():
    lib = _capi.test_lib()
    V = [({}, -1), ({"precision": 3}, -1), ({"precision": 3, "overlap": 41 | 4}, -1), ({"precision": 3, "gemm_persistent": 7}, -1)]
    for (H, W, Cin, Cout, KS, dil, resid, act) in [(64, 128, 512, 512, 3, 4, True, 1),      # layer 4's conv: 36 GEMMs of K = 512, even dilation -> row-parity chunks
                                                   (64, 128, 256, 256, 3, 2, False, 0),     # layer 3
                                                   (33, 65, 128, 132, 3, 1, True, 1),       # ragged tile rows, N not a multiple of 128
                                                   (97, 193, 512, 512, 1, 1, False, 0),     # the Encoding's value conv at the native feature size
                                                   (40, 52, 512, 64, 1, 1, True, 2),        # 512 -> 64 (half a tile of padding), residual, LeakyReLU
                                                   (17, 33, 2048, 512, 1, 1, False, 1),     # K = 2048 (a Bottleneck conv1): 128 steps
                                                   (97, 193, 1024, 256, 1, 1, False, 1),    # psp101 layer3's conv1 at the native size: 148 tiles, on the split GEMM by the deep-K rule
                                                   (128, 256, 64, 64, 3, 1, True, 1),       # ResNet layer1 (k_conv_adirect_b3, fusion bit 524288)
                                                   (37, 53, 96, 48, 3, 3, False, 0)]:       # the same kernel: 27 steps, ragged rows and channels, dilation 3
        vs = [v for v in V if not ((v[0].get("overlap", 0) & 1) and (KS == 1 or dil % 2))]
        if Cout <= 64 and KS == 3:
            vs = [({}, -1), ({"precision": 2}, -1)]                    # fusion bit 524288 (default): the split kernel whatever the size
        e = _conv_errors(lib, H, W, Cin, Cout, KS, dil, resid, act if act != 2 else 0, vs)
        print("conv %dx%d %d->%d k%d d%d: fp32 max %.2e rms %.2e | split %s" % (H, W, Cin, Cout, KS, dil, e[0][0], e[0][1], " ".join("max %.2e rms %.2e" % x for x in e[1:])))
        for x in e[1:]:
            assert x[0] <= 1.25 * e[0][0] + 1e-7 and x[1] <= 1.1 * e[0][1] + 1e-9, (H, W, Cin, Cout, KS, dil, e)
    # and against the plain fp32 torch reference with the tolerance of the fp32 operator tests
    mem = opcheck.TorchMem()
    opcheck.conv(lib, mem, 13, 21, 128, 128, 3, 1, 2, 1, True, opts={"precision": 3})
    opcheck.conv(lib, mem, 11, 19, 64, 160, 1, 1, 1, 2, True, opts={"precision": 3})
    opcheck.conv(lib, mem, 300, 40, 64, 256, 1, 1, 1, 0, False, opts={"precision": 3, "gemm_persistent": 11})
    opcheck.conv(lib, mem, 61, 77, 64, 64, 3, 1, 1, 1, True, opts={"precision": 2})            # k_conv_adirect_b3: layer1, ragged rows
    opcheck.conv(lib, mem, 40, 52, 96, 40, 3, 2, 2, 2, False, opts={"precision": 2})           # stride 2, dilation 2, 27 steps (odd), ragged channels
    opcheck.conv(lib, mem, 33, 65, 64, 64, 1, 2, 1, 0, False, opts={"precision": 2})           # the strided 1x1 form
    opcheck.conv(lib, mem, 129, 257, 64, 128, 3, 2, 1, 1, False, opts={"precision": 2})        # 65 .. 128 output channels as two 64-column tiles: layer2.0's strided conv
    opcheck.conv(lib, mem, 97, 193, 64, 128, 3, 1, 1, 1, False, opts={"precision": 2, "winograd": 0})   # a deep stem's 64 -> 128 conv (Cin < 128: never Winograd by default either)
    opcheck.conv(lib, mem, 40, 52, 64, 100, 1, 2, 1, 0, True, opts={"precision": 2})           # strided 1x1 downsample with a residual, ragged second tile
    opcheck.conv(lib, mem, 97, 193, 256, 64, 1, 1, 1, 1, False, opts={"precision": 2})         # a Bottleneck's conv1 in layer1 (stride-1 1x1, 18721 pixels): the split direct kernel, not the fp32 GEMM
    opcheck.conv(lib, mem, 97, 193, 512, 128, 1, 1, 1, 1, True, opts={"precision": 2})         # layer2's conv1: two 64-column tiles, 16 K steps
    for H, W in ((131, 259), (224, 224), (7, 9)):                                               # the packed-row 7x7 stem on the split kernel
        opcheck.stem(lib, mem, H, W, opts={"precision": 2})
    # the new kernels really ran (their sums are not the fp32 MFMA's bit for bit) and a handle without the bit keeps the exact-fp32 ones
    g = np.random.default_rng(3)
    x = torch.from_numpy(g.standard_normal((61, 77, 64)).astype(np.float32)).cuda()
    w = (g.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32)
    outs = []
    for kw in ({}, {"precision": 2}, {"precision": 2, "fusion": lib.opts().fusion & ~524288}):
        out = torch.empty(61, 77, 64, device="cuda")
        o = lib.opts(**kw)
        lib.check(lib.tdnet_op_conv2d(x.data_ptr(), 61, 77, 64, w.ctypes.data, None, 64, 3, 1, 1, None, 0, ctypes.byref(o), -1, out.data_ptr(), None))
        outs.append(out.cpu())
    assert not torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and (outs[0] - outs[1]).abs().max() < 1e-4


def test_split_attention_is_as_accurate_as_the_fp32_kernel():
    """td_attn_b3.h (tdnet_op_attention online = 17) against an fp64 evaluation at the frame's shapes and at ragged ones, next to the exact-fp32 kernel
    (online = 2) on the same inputs: softmax(q k^T / 8) v' + bias + resid with |scores| up to ~30."""
    lib = _capi.test_lib()
    g = torch.Generator(device="cpu").manual_seed(1)
    s = torch.cuda.current_stream().cuda_stream
    for (Lq, Lk, DV, scale) in ((32768, 2048, 512, 0.5), (18721, 1225, 512, 0.5), (32768, 2048, 128, 0.5), (4099, 777, 512, 2.0), (333, 45, 128, 1.0)):
        q = (torch.randn(Lq, 64, generator=g) * scale).cuda(); k = (torch.randn(Lk, 64, generator=g) * scale).cuda()
        vp = torch.zeros((Lk + 127) // 128 * 128, DV); vp[:Lk] = torch.randn(Lk, DV, generator=g); vp = vp.cuda()
        b = torch.randn(DV, generator=g).cuda(); r = torch.randn(Lq, DV, generator=g).cuda()
        ref = torch.softmax(q.double() @ k.double().t() / 8.0, dim=1) @ vp[:Lk].double() + b.double() + r.double()
        errs = []
        for online in (2 | 32, 17) + ((18, 19) if DV == 512 else ()):        # 17: the form a frame picks for this Lq; 18 / 19: the 64- / 32-query form (DV = 512 k)
            out = torch.full((Lq, DV), 7e7, device="cuda")
            lib.check(lib.tdnet_op_attention(q.data_ptr(), k.data_ptr(), vp.data_ptr(), b.data_ptr(), r.data_ptr(), Lq, Lk, DV, online, None, None, None, out.data_ptr(), s))
            e = (out.double() - ref).abs()
            errs.append((e.max().item(), e.pow(2).mean().sqrt().item()))
        print("attention Lq %d Lk %d DV %d: fp32 max %.2e rms %.2e | split %s" % (Lq, Lk, DV, errs[0][0], errs[0][1], " | ".join("max %.2e rms %.2e" % x for x in errs[1:])))
        for x in errs[1:]:
            assert x[0] <= 1.25 * errs[0][0] + 1e-7 and x[1] <= 1.1 * errs[0][1] + 1e-9, (Lq, Lk, DV, errs)
    mem = opcheck.TorchMem()
    opcheck.attention(lib, mem, 300, 200, 512, spike=True, online=17, ln=True)       # + the plane LayerNorm from the epilogue's strip statistics
    opcheck.attention(lib, mem, 130, 193, 128, True, True, online=17, ln=True)
    opcheck.attention(lib, mem, 18721, 1225, 512, online=18, ln=True)                # the eight-wave form at the native feature size: 293 workgroups, the last one 33 queries
    opcheck.attention(lib, mem, 300, 200, 512, spike=True, online=18, ln=True)


def test_precision2_meets_the_fp32_gate_1024x2048():
    _vs_oracle("td4", "resnet18", 1024, 2048, 5, kernel_opts={"precision": 2})         # configs[2]: four cold paths + the first steady-state frame


def test_precision2_meets_the_fp32_gate_native_769x1537_and_td2():
    _vs_oracle("td4", "resnet18", 769, 1537, 5, kernel_opts={"precision": 2})
    _vs_oracle("td2", "resnet18", 1024, 2048, 3, kernel_opts={"precision": 2})         # configs[1]


def test_precision2_bottleneck_backbone_with_the_narrow_conv_routings():
    """td2-psp50 at 385x769: the deep stem's 64 -> 128 conv as two column tiles of the split direct kernel, layer1's stride-1 1x1 convs to 64 channels (18721
    pixels: above the 8192-pixel rule) on k_conv_adirect_b3<1> instead of the fp32 GEMM, the wide 1x1 convs on the split GEMM where its size rule lets them."""
    _vs_oracle("td2", "resnet50", 385, 769, 3, kernel_opts={"precision": 2})


def test_precision2_split_kernel_forced_on_small_maps():
    _vs_oracle("td4", "resnet18", 257, 513, 8, kernel_opts={"precision": 3})           # every path cold and in steady state; ragged 256-row tiles everywhere
    _vs_oracle("td2", "resnet50", 129, 257, 4, kernel_opts={"precision": 3})           # Bottleneck backbone: the 1x1 convs (K up to 2048) on the split kernel
    _vs_oracle("td4", "resnet18", 257, 513, 5, kernel_opts={"precision": 3, "overlap": 41 | 4})   # + row-parity chains
    _vs_oracle("td4", "resnet50", 129, 257, 6, kernel_opts={"precision": 3})           # d_v = 2048: the split attention as four 512-channel launches on one pre-split V'


def test_precision2_psp101_small_golden(golden_dir):
    """The stateless comparison model (pspnet.py:31-115) with every eligible GEMM on the split kernel -- the Bottleneck 1x1 convs of ResNet-101 with K up to
    2048, the Winograd convs, the head -- against the golden captured from the real reference."""
    import os
    from tdnet_amd import weights
    from tdnet_amd.model import pspnet
    g = np.load(os.path.join(golden_dir, "psp_resnet101_33x65.npz"))
    m = pspnet.pspnet(nclass=19, model_path=None, synthetic_seed=0, kernel_opts={"precision": 3}).eval().to("cuda")
    x = torch.from_numpy(weights.synth_video(33, 65, 1, seed=1)[0]).cuda()
    with torch.no_grad():
        out = m(x, pos_id=0).cpu().numpy()
    assert m.engine.opts()["precision"] == 3
    assert np.abs(out - g["f0_logits"]).max() <= 1e-3 and (out[0].argmax(0) == g["f0_logits"][0].argmax(0)).all()


def test_precision2_uncalibrated_reference_init_129x257():
    """test_gpu_model.test_uncalibrated_reference_init's protocol (three clips pooled; 4x max / 3x rms of the CPU path's own error; every clip
    <= 5e-4 max|truth|) with the split kernel forced at this small size, next to the exact-fp32 kernels on the same clips."""
    rows = {}
    for tag, extra in (("exact fp32", None), ("bf16x3", {"precision": 3})):
        runs = [_reference_init_stress(129, 257, 5, 3, seed=seed, gate=False, extra_opts=extra) for seed in (1, 2, 3)]
        e_gpu, e_cpu = max(r["e_gpu"] for r in runs), max(r["e_cpu"] for r in runs)
        n = sum(r["n"] for r in runs)
        r_gpu, r_cpu = (sum(r["s_gpu"] for r in runs) / n) ** 0.5, (sum(r["s_cpu"] for r in runs) / n) ** 0.5
        rows[tag] = (e_gpu, e_cpu, r_gpu, r_cpu, runs)
        print("129x257 pooled over 3 clips, %s: e_gpu/e_cpu max %.2e / %.2e = x%.2f, rms %.2e / %.2e = x%.2f" % (tag, e_gpu, e_cpu, e_gpu / e_cpu, r_gpu, r_cpu, r_gpu / r_cpu))
    e_gpu, e_cpu, r_gpu, r_cpu, runs = rows["bf16x3"]
    assert e_gpu <= 4.0 * e_cpu and r_gpu <= 3.0 * r_cpu, rows["bf16x3"][:4]
    for r in runs:
        assert r["e_gpu"] <= 5e-4 * r["tmax"], r


def test_precision2_uncalibrated_reference_init_769x1537():
    """The checkpoint's geometry ([97,193] LayerNorm affine), un-calibrated init, P + 2 frames: the gate of
    test_gpu_model.test_reference_init_at_the_checkpoints_geometry_769x1537 (4x / 3x, max|gpu - cpu| <= 3x the CPU's own error) in precision 2."""
    r = _reference_init_stress(769, 1537, 6, 3, gc_bound=3.0, extra_opts={"precision": 2})
    print("769x1537 bf16x3: e_gpu/e_cpu x%.2f (max), x%.2f (rms)" % (r["e_gpu"] / r["e_cpu"], (r["s_gpu"] / r["s_cpu"]) ** 0.5))
