"""The CPU oracle (oracle/tdnet_ref.py) against golden vectors captured from the real reference
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tdnet_ref
from tdnet_amd import arch, weights

CASES = [("td4", "resnet18", 33, 65), ("td2", "resnet18", 33, 65), ("td2", "resnet34", 33, 65),
         ("td4", "resnet18", 65, 129), ("td2", "resnet18", 49, 81), ("td2", "resnet50", 33, 65), ("td4", "resnet34", 33, 65),
         ("td4", "resnet50", 33, 65)]


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
    T = int(g[tag + "_last_frame"]) + 1
    outs = _run("td2", "resnet18", 512, 1024, T)
    checked = 0
    for t in range(T):                                                  # every steady-state frame has its own digest
        if "%s_f%d_sample" % (tag, t) not in g.files:
            continue
        out = outs[t]["logits"]
        assert np.abs(out[0, :, ::61, ::67] - g["%s_f%d_sample" % (tag, t)]).max() <= 1e-3
        stats = np.array([out.min(), out.max(), out.mean(), np.sqrt((out.astype(np.float64) ** 2).sum())])
        assert np.allclose(stats, g["%s_f%d_stats" % (tag, t)], rtol=1e-4, atol=1e-4)
        assert (out[0].argmax(0)[::61, ::67] != g["%s_f%d_labels_sample" % (tag, t)]).mean() <= 0.002
        checked += 1
    assert checked == T - 1
    assert np.array_equal(g[tag + "_sample"], g["%s_f%d_sample" % (tag, T - 1)])


def test_goldens_cover_every_path_in_steady_state(golden_dir):
    """The fixtures from the real reference hold every sub-network's steady-state frame at least twice: td4 needs t >= 3 and
    pos_id = t mod 4 in {0,1,2,3} (forward_path3 = atn3_4 -> atn3_1 -> atn3_2, td4_psp18.py:176-195, first at t = 6)."""
    for fn, P, fifo in [("td4_resnet18_33x65.npz", 4, 3), ("td4_resnet18_65x129.npz", 4, 3), ("td4_resnet34_33x65.npz", 4, 3), ("td4_resnet50_33x65.npz", 4, 3),
                        ("td2_resnet18_33x65.npz", 2, 1), ("td2_resnet34_33x65.npz", 2, 1), ("td2_resnet50_33x65.npz", 2, 1)]:
        g = np.load(os.path.join(golden_dir, fn))
        T = 1 + max(int(k.split("_")[0][1:]) for k in g.files if k.startswith("f"))
        seen = {}
        for t in range(fifo, T):
            seen[t % P] = seen.get(t % P, 0) + 1
        assert sorted(seen) == list(range(P)) and min(seen.values()) >= (1 if "resnet50_33" in fn and P == 4 else 2), (fn, seen)
    d = np.load(os.path.join(golden_dir, "fullsize_digests.npz"))
    for tag, P, fifo in [("td4_resnet18_1024x2048", 4, 3), ("td4_resnet18_769x1537", 4, 3), ("td2_resnet18_1024x2048", 2, 1)]:
        T = int(d[tag + "_last_frame"]) + 1
        assert {t % P for t in range(fifo, T) if "%s_f%d_sample" % (tag, t) in d.files} == set(range(P)), tag


def test_oracle_architecture_facts_agree_with_the_product_side():
    """oracle/tdnet_ref.py restates block list, attention order, pyramid slices and FIFO depth from the reference on its own; the
    HIP side reads tdnet_amd/arch.py.  Two independent statements of the same reference lines must agree (a disagreement would
    otherwise show up as a parity failure with no hint where)."""
    for bb in ("resnet18", "resnet34", "resnet50", "resnet101"):
        a, b = arch.backbone_blocks(bb), tdnet_ref.ref_backbone_blocks(bb)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x.name, x.kind, x.stride, x.dil1, x.dil2, x.downsample) == (y.name, y.kind, y.stride, y.dil1, y.dil2, y.downsample)
    for name in ("td4", "td2"):
        spec = arch.model_spec(name, 19, "resnet18")
        assert tuple(spec.atn_names[p] for p in range(spec.path_num)) == tdnet_ref.REF_ATN_ORDER[name]
        assert (spec.psp_path_num, tuple(spec.pids)) == tdnet_ref.REF_PSP[name] and spec.fifo == tdnet_ref.REF_FIFO[name]


def test_state_dict_shape_inventory():
    """td4 must expose the 728-tensor / 54.83 M-parameter inventory of the reference checkpoint."""
    spec = arch.model_spec("td4")
    shapes = arch.state_dict_shapes(spec, 97, 193)
    assert len(shapes) == 728
    n = sum(int(np.prod(s)) for k, s in shapes.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert abs(n / 1e6 - 54.83) < 0.1
