"""The plain-C restatement of the L0 operators (oracle/ops_c.c, bound by oracle/c_ops.py): each operator against PyTorch's on
ragged shapes, then the WHOLE oracle graph on the C operators -- not one PyTorch kernel in the loop -- against the golden
vectors captured from the real reference (tests/golden/, tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_ops, tdnet_ref
from tdnet_amd import arch, weights

C = c_ops.COps


def _close(a, b, tol):
    err = float((a.double() - b.double()).abs().max())
    assert err <= tol, err


def test_operators_against_pytorch():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    for (cin, cout, h, w, ks, st, dil) in [(5, 7, 13, 21, 3, 1, 1), (4, 6, 12, 30, 3, 1, 4), (3, 8, 33, 65, 7, 2, 1), (6, 4, 9, 17, 3, 2, 1),
                                           (8, 5, 7, 9, 1, 1, 1), (8, 5, 11, 9, 1, 4, 1), (4, 4, 5, 9, 3, 1, 16)]:
        x, wt, b = r(1, cin, h, w), r(cout, cin, ks, ks), r(cout)
        pad = 3 if ks == 7 else dil * (ks // 2)
        _close(C.conv2d(x, wt, b, st, pad, dil), F.conv2d(x, wt, b, st, pad, dil), 2e-5)
        _close(C.conv2d(x, wt, None, st, pad, dil), F.conv2d(x, wt, None, st, pad, dil), 2e-5)
    x = r(1, 6, 17, 33)
    assert torch.equal(C.max_pool2d(x, 3, 2, 1), F.max_pool2d(x, 3, 2, 1))
    assert torch.equal(C.relu(x), F.relu(x)) and torch.equal(C.leaky_relu(x, 0.01), F.leaky_relu(x, 0.01))
    x = r(1, 5, 97 // 4, 193 // 4 + 1)                                  # non-divisible sizes: overlapping bins
    for o in (1, 2, 3, 6):
        _close(C.adaptive_avg_pool2d(x, o), F.adaptive_avg_pool2d(x, o), 1e-6)
    for (hi, wi, ho, wo) in [(6, 6, 97, 193), (1, 1, 13, 7), (5, 9, 33, 65), (3, 2, 3, 2)]:
        x = r(1, 4, hi, wi)
        _close(C.interpolate(x, (ho, wo), mode="bilinear", align_corners=True), F.interpolate(x, (ho, wo), mode="bilinear", align_corners=True), 1e-5)
    q, k, v = r(1, 45, 64), r(1, 23, 64), r(1, 23, 40)
    s = C.bmm(q, k.transpose(1, 2))
    _close(s, torch.bmm(q, k.transpose(1, 2)), 2e-5)
    p = C.softmax(s / 8.0, dim=2)
    _close(p, torch.softmax(s / 8.0, dim=2), 1e-6)
    _close(C.bmm(p, v), torch.bmm(p, v), 1e-5)
    wm = r(40, 40)
    _close(C.matmul(C.bmm(p, v), wm.t()), torch.matmul(torch.bmm(p, v), wm.t()), 2e-5)
    x, gw, gb = r(1, 6, 5, 9), r(5, 9), r(5, 9)
    _close(C.layer_norm(x, (5, 9), gw, gb, 1e-5), F.layer_norm(x, (5, 9), gw, gb, 1e-5), 1e-5)


@pytest.mark.parametrize("name,bb", [("td4", "resnet18"), ("td2", "resnet18"), ("td2", "resnet34"), ("td2", "resnet50")])
def test_graph_on_c_operators_matches_the_reference_goldens(golden_dir, name, bb):
    H, W = 33, 65
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    T = 1 + max(int(k.split("_")[0][1:]) for k in g.files if k.startswith("f"))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.synth_state_dict(spec, h, w, 0).items()}
    prev = tdnet_ref.set_ops(C)
    try:
        ref = tdnet_ref.TDNetRef(spec, sd)
        for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
            ref.trace = {}
            out = ref.forward(torch.from_numpy(x), t % spec.path_num).numpy()
            for st in ("c4", "z", "v_cur", "ln", "lowres"):
                if "f%d_%s" % (t, st) not in g.files:
                    continue
                gold = g["f%d_%s" % (t, st)]
                got = ref.trace[st].numpy()
                assert np.abs(got - gold).max() <= 1e-4 * max(1.0, np.abs(gold).max()), (t, st)
            gold = g["f%d_logits" % t]
            assert np.abs(out - gold).max() <= 1e-4, t
            flips = int((out[0].argmax(0) != gold[0].argmax(0)).sum())
            assert flips <= 2, (t, flips)                                       # ties at the 1e-6 level only
    finally:
        tdnet_ref.set_ops(prev)


def test_pspnet101_graph_on_c_operators_matches_the_reference_golden(golden_dir):
    """The stateless comparison model (deep stem, Bottleneck blocks, full pyramid pooling) on the C operators."""
    H, W = 33, 65
    spec = arch.model_spec("psp", 19, "resnet101")
    h, w = arch.feat_size(H), arch.feat_size(W)
    g = np.load(os.path.join(golden_dir, "psp_resnet101_%dx%d.npz" % (H, W)))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.synth_state_dict(spec, h, w, 0).items()}
    prev = tdnet_ref.set_ops(C)
    try:
        ref = tdnet_ref.PSPNetRef(spec, sd)
        frames = sorted(int(k.split("_")[0][1:]) for k in g.files if k.endswith("_logits"))
        for t, x in zip(frames, weights.synth_video(H, W, len(frames), seed=1)):
            out = ref.forward(torch.from_numpy(x)).numpy()
            gold = g["f%d_logits" % t]
            assert np.abs(out - gold).max() <= 2e-4 * max(1.0, np.abs(gold).max()), t
    finally:
        tdnet_ref.set_ops(prev)
