"""The single-frame PSPNet-101 comparison model (Testing/model/pspnet/pspnet.py, `--model psp101`): oracle vs goldens from
the real reference, the HIP sources under the emulator, and (gpu) the kernels themselves."""
import os

import numpy as np
import pytest
import torch

import emu_util
from oracle import tdnet_ref
from tdnet_amd import arch, weights
from tdnet_amd.engine import Engine

H, W = 33, 65


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "psp_resnet101_%dx%d.npz" % (H, W)))


def test_oracle_matches_reference(golden_dir):
    g = _golden(golden_dir)
    spec = arch.model_spec("psp", 19, "resnet101")
    ref = tdnet_ref.PSPNetRef(spec, weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0))
    ref.trace = {}
    out = ref.forward(torch.from_numpy(weights.synth_video(H, W, 1, seed=1)[0])).numpy()
    for k in ("c4", "z", "lowres"):
        assert np.abs(ref.trace[k].numpy() - g["f0_" + k]).max() <= 1e-4 * max(1.0, np.abs(g["f0_" + k]).max()), k
    assert np.abs(out - g["f0_logits"]).max() <= 1e-3


def test_emulated_kernels_match_reference(golden_dir):
    g = _golden(golden_dir)
    spec = arch.model_spec("psp", 19, "resnet101")
    h, w = arch.feat_size(H), arch.feat_size(W)
    e = Engine(1, 101, 19, H, W, 0, lib=emu_util.emu_lib())
    e.load_state_dict(weights.synth_state_dict(spec, h, w, 0))
    out = np.full((1, 19, H, W), 7e7, np.float32)
    e.forward(weights.synth_video(H, W, 1, seed=1)[0], 0, out)
    assert e.fifo_len() == 0
    for k, shp in (("c4", (1, 2048, h, w)), ("z", (1, 4096, h, w)), ("lowres", (1, 19, h, w))):
        assert np.abs(e.stage(k, shp) - g["f0_" + k]).max() <= 1e-4 * max(1.0, np.abs(g["f0_" + k]).max()), k
    assert np.abs(out - g["f0_logits"]).max() <= 1e-3
    assert (out[0].argmax(0) == g["f0_logits"][0].argmax(0)).all()
    e.close()


@pytest.mark.gpu
def test_gpu_psp101(golden_dir):
    from tdnet_amd.model import pspnet
    g = _golden(golden_dir)
    m = pspnet.pspnet(nclass=19, model_path=None, synthetic_seed=0).eval().to("cuda")
    x = torch.from_numpy(weights.synth_video(H, W, 1, seed=1)[0]).cuda()
    with torch.no_grad():
        out = m(x, pos_id=0).cpu().numpy()
        again = m(x).cpu().numpy()                                   # stateless: pos_id ignored, same result
    assert np.abs(out - g["f0_logits"]).max() <= 1e-3 and np.array_equal(out, again)
    # full size against the reference's digest and the oracle (769x1537, the size test.py:36 uses)
    d = np.load(os.path.join(golden_dir, "fullsize_digests.npz"))
    Hf, Wf = 769, 1537
    spec = arch.model_spec("psp", 19, "resnet101")
    m2 = pspnet.pspnet(nclass=19, model_path=None, synthetic_seed=0).eval().to("cuda")
    xf = weights.synth_video(Hf, Wf, 1, seed=1)[0]
    with torch.no_grad():
        o = m2(torch.from_numpy(xf).cuda()).cpu().numpy()
    assert np.abs(o[0, :, ::61, ::67] - d["psp_resnet101_769x1537_sample"]).max() <= 2e-3     # |logits| reach 42 here
    assert (o[0].argmax(0)[::61, ::67] != d["psp_resnet101_769x1537_labels_sample"]).mean() <= 0.002
    tdnet_ref.tune_threads()
    ref = tdnet_ref.PSPNetRef(spec, weights.synth_state_dict(spec, arch.feat_size(Hf), arch.feat_size(Wf), 0))
    exp = ref.forward(torch.from_numpy(xf)).numpy()
    err = float(np.abs(o - exp).max())
    bad = o[0].argmax(0) != exp[0].argmax(0)
    top2 = np.sort(exp[0], axis=0)[-2:]
    print("psp101 769x1537: max|dlogit| %.2e (|logit| max %.1f), %d label flips" % (err, np.abs(exp).max(), int(bad.sum())))
    assert err <= 2e-3 and ((top2[1] - top2[0])[bad] <= 2 * err).all()
