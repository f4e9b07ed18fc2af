"""The CPU oracle (oracle/tdnet_ref.py) against golden vectors captured from the real reference
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tdnet_ref
from tdnet_amd import arch, weights

CASES = [("td4", "resnet18", 33, 65), ("td2", "resnet18", 33, 65), ("td2", "resnet34", 33, 65),
         ("td4", "resnet18", 65, 129), ("td2", "resnet18", 49, 81), ("td2", "resnet50", 33, 65)]


def _run(name, bb, H, W, T):
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    net = tdnet_ref.TDNetRef(spec, weights.synth_state_dict(spec, h, w, 0))
    frames = weights.synth_video(H, W, T, seed=1)
    outs = []
    for t, x in enumerate(frames):
        net.trace = {}
        out = net.forward(torch.from_numpy(x), t % spec.path_num)
        rec = {k: v.numpy() for k, v in net.trace.items()}
        rec["logits"] = out.numpy()
        outs.append(rec)
    return outs


@pytest.mark.parametrize("name,bb,H,W", CASES)
def test_oracle_matches_reference_goldens(golden_dir, name, bb, H, W):
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    T = 1 + max(int(k.split("_")[0][1:]) for k in g.files if k.startswith("f"))
    outs = _run(name, bb, H, W, T)
    checked = 0
    for key in g.files:
        if not key.startswith("f"):
            continue
        t, stage = key.split("_", 1)
        got = outs[int(t[1:])][stage]
        ref = g[key]
        assert got.shape == ref.shape, key
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 1e-4 * scale, (key, np.abs(got - ref).max(), scale)
        checked += 1
    assert checked >= T * 4
    # argmax identity on the final logits (tie-tolerant: a flip is only allowed inside the reference's top-2 gap)
    for t in range(T):
        ref = g["f%d_logits" % t][0]
        got = outs[t]["logits"][0]
        bad = ref.argmax(0) != got.argmax(0)
        if bad.any():
            top2 = np.sort(ref, axis=0)[-2:]
            assert ((top2[1] - top2[0])[bad] <= 1e-4).all()


def test_oracle_fullsize_c1_digest(golden_dir):
    """BASELINE config[0]: td2-psp18, 512x1024, 4-frame clip, CPU."""
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "fullsize_digests.npz"))
    tag = "td2_resnet18_512x1024"
    outs = _run("td2", "resnet18", 512, 1024, int(g[tag + "_last_frame"]) + 1)
    out = outs[-1]["logits"]
    samp = out[0, :, ::61, ::67]
    assert np.abs(samp - g[tag + "_sample"]).max() <= 1e-3
    stats = np.array([out.min(), out.max(), out.mean(), np.sqrt((out.astype(np.float64) ** 2).sum())])
    assert np.allclose(stats, g[tag + "_stats"], rtol=1e-4, atol=1e-4)
    lab = out[0].argmax(0)[::61, ::67]
    assert (lab != g[tag + "_labels_sample"]).mean() <= 0.002


def test_state_dict_shape_inventory():
    """td4 must expose the 728-tensor / 54.83 M-parameter inventory of the reference checkpoint."""
    spec = arch.model_spec("td4")
    shapes = arch.state_dict_shapes(spec, 97, 193)
    assert len(shapes) == 728
    n = sum(int(np.prod(s)) for k, s in shapes.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert abs(n / 1e6 - 54.83) < 0.1
