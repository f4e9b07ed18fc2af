"""Synthetic, reproducible weights with the reference's state_dict names and shapes.

The reference's checkpoints are not available (Testing/TEST_README.md:7-8 points at Google Drive), and a
missing file silently leaves random init (td4_psp18.py:239-240).  Parity is therefore checked on seeded
synthetic weights that BOTH sides can regenerate bit-identically without shipping hundreds of MB:
a counter-based generator (numpy Philox) keyed by (seed, crc32(tensor name)).

BN statistics/affine, conv biases and the LayerNorm affine are randomised on purpose: with default
(identity-like) BN a wrong BN fold would go unnoticed (SURVEY.md §7 "hard parts").
"""
import re
import zlib

import numpy as np

from . import arch


# Gain on the second (output) conv of the q and k branches, calibrated once with the CPU oracle so that the
# attention scores q.k^T/8 have an rms of ~2-3 at 1024x2048 (softmax neither uniform nor one-hot).
QK_GAIN = 0.3
BOTTLENECK_GAIN = 1.0
BOTTLENECK_DEPTH_EXP = 0.54      # ResNet-101 (33 blocks) vs ResNet-50 (16): keeps c4 at the same scale


def _is_qk_out(name):
    return ".w_qs.1." in name or ".w_ks.1." in name


def _is_linear_conv(name):
    return ".w_vs.0." in name or ".fc.0." in name or _is_qk_out(name) or ".conv5.4." in name or name.startswith("head.conv5.5.")


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def synth_tensor(name, shape, seed, init="calibrated"):
    """init = "calibrated" (default, see below) or "reference": SURVEY.md 8d's original recipe -- every conv ~ N(0, 2/(k*k*C_out)) as
    resnet.py:162-165 initialises its convs, no q/k gain, no depth normalisation.  With it activations grow ~8x through the 512->64
    projections and the attention scores reach the hundreds; it exists to STRESS the kernels (relative-error gate,
    tests/test_gpu_model.py::test_uncalibrated_reference_init), not to pin absolute logits."""
    g = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros((), dtype=np.int64)
    if ".fc." in name and name.startswith("pretrained"):       # also "pretrained.fc.*" of pspnet
        return np.zeros(shape, dtype=np.float32)            # unused classifier (resnet.py:159-160), kept for strict load
    if ".ln." in name:
        if leaf == "weight":
            return g.uniform(0.5, 1.5, shape).astype(np.float32)
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4 and init == "reference":
        return (np.sqrt(2.0 / (shape[2] * shape[3] * shape[0])) * g.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4:
        # conv weight: He-style normal as resnet.py:162-165, but with n = k*k*max(C_in, C_out) instead of the
        # reference's k*k*C_out: fan-out scaling blows activations up 8x through the 512->64 projections, which
        # makes q.k^T/8 reach the hundreds and the softmax so peaked that two fp32 CPU evaluations of the SAME
        # graph differ by >1e-3 in the logits.  Trained weights do not behave like that; this keeps every stage O(1).
        n = shape[2] * shape[3] * max(shape[0], shape[1])
        gain = 1.0 if _is_linear_conv(name) else 2.0        # no ReLU behind w_vs / fc / second q,k conv / classifier
        wt = (np.sqrt(gain / n) * g.standard_normal(shape)).astype(np.float32)
        return wt * np.float32(QK_GAIN) if _is_qk_out(name) else wt
    if len(shape) == 1 and leaf == "weight":                # BN gamma
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "bias" and _is_qk_out(name) and init != "reference":
        return (0.05 * QK_GAIN * g.standard_normal(shape)).astype(np.float32)
    if leaf == "bias":
        is_bn = name.endswith(("bn1.bias", "bn2.bias", "bn3.bias", "bn.bias", "conv1.1.bias", "conv1.4.bias")) or ".downsample.1." in name \
            or (".conv5.1." in name) or (".conv5.2." in name) or (name.startswith("psp") and ".1." in name) \
            or (name.startswith("head.conv5.0.conv") and ".1." in name)
        return ((0.1 if is_bn else 0.05) * g.standard_normal(shape)).astype(np.float32)
    raise ValueError("no rule for %s %s" % (name, shape))


_BN_LAST = {False: re.compile(r"^pretrained\d*\.layer\d\.\d+\.bn2\.weight$"),      # BasicBlock: bn2 closes the branch
            True: re.compile(r"^pretrained\d*\.layer\d\.\d+\.bn3\.weight$")}       # Bottleneck: bn3


def synth_state_dict(spec, h, w, seed=0, init="calibrated"):
    """{name: np.ndarray} with exactly the reference's keys for `spec` at feature size h x w.

    The gamma of every residual branch's last BN is scaled by (8 / n_blocks)^0.77: each BasicBlock adds its branch
    variance to the trunk, so an un-normalised 16-block ResNet-34 ends ~2x hotter than the 8-block ResNet-18, the
    attention scores grow with it and fp32 evaluations of the same graph drift apart (see QK_GAIN above).  The factor
    is 1 for ResNet-18."""
    nblocks = len(arch.backbone_blocks(spec.backbone))
    # measured with the CPU oracle: these factors keep c4 rms of ResNet-34 / ResNet-50 at the ResNet-18 level (~9 at full size)
    g2 = np.float32(BOTTLENECK_GAIN * (16.0 / nblocks) ** BOTTLENECK_DEPTH_EXP if arch.is_bottleneck(spec.backbone) else (8.0 / nblocks) ** 0.77)
    if init == "reference":
        g2 = np.float32(1.0)
    out = {}
    for k, s in arch.state_dict_shapes(spec, h, w).items():
        t = synth_tensor(k, s, seed, init)
        if g2 != 1.0 and _BN_LAST[arch.is_bottleneck(spec.backbone)].match(k):
            t = t * g2
        out[k] = t
    return out


def synth_video(H, W, n_frames, seed=0):
    """Synthetic clip in the loader's output range (dataloader.py:66-73): x_t = clip(base + 0.02 t drift)."""
    g = _rng(seed, "video/%dx%d" % (H, W))
    base = g.standard_normal((1, 3, H, W)).astype(np.float32)
    drift = g.standard_normal((1, 3, H, W)).astype(np.float32)
    return [np.clip(base + np.float32(0.02 * t) * drift, -2.2, 2.7).astype(np.float32) for t in range(n_frames)]
