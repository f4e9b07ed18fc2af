"""GPU parity, model level: the Python model classes (reference API) -> C ABI -> HIP kernels, against
 (a) the golden vectors captured from the real reference (small sizes, every stage),
 (b) the CPU oracle on the same seeded inputs at mid and FULL (1024x2048, 769x1537) size,
 (c) size-independent properties (determinism, forward_labels == argmax(forward), reset, FIFO semantics).
Gate (SURVEY.md §7 / BASELINE.md §3): max|dlogit| <= 1e-3; label flips only where the reference's top-2 gap is within
2*max|dlogit|; mIoU(pred, ref_pred) >= 0.9995."""
import os

import numpy as np
import pytest
import torch

from oracle import tdnet_ref
from tdnet_amd import arch, weights
from tdnet_amd.model import td2_psp50, td4_psp18

pytestmark = pytest.mark.gpu


def make_model(name, bb, seed=0, kernel_opts=None):
    if name == "td4":
        return td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, backbone=bb, synthetic_seed=seed, kernel_opts=kernel_opts).eval().to("cuda")
    return td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=None, backbone=bb, synthetic_seed=seed, kernel_opts=kernel_opts).eval().to("cuda")


def check_frame(out, ref, tag, hist=None):
    """Per frame: max|dlogit| <= 1e-3 and label flips only inside the reference's top-2 tie band.  mIoU(pred, ref_pred) >= 0.9995 is a
    property of the STREAM (confusion matrix over the benchmark clip): with `hist` the frame's confusion matrix is added
    to it and the caller gates the clip; without, the frame is gated on its own (full-size frames only -- at 257x513 one tie-band
    flip in a class of a few hundred pixels moves a single frame's mIoU by more than 5e-4)."""
    err = float(np.abs(out - ref).max())
    assert err <= 1e-3, (tag, err)
    lo, lr = out[0].argmax(0), ref[0].argmax(0)
    bad = lo != lr
    if bad.any():
        top2 = np.sort(ref[0], axis=0)[-2:]
        assert ((top2[1] - top2[0])[bad] <= 2 * err).all(), (tag, "label flip outside the tie band")
    miou, h = tdnet_ref.confusion_miou(lo, lr, 19)
    if hist is not None:
        hist += h
    else:
        assert miou >= 0.9995, (tag, miou)
    return err, int(bad.sum())


def clip_miou(hist):
    with np.errstate(divide="ignore", invalid="ignore"):
        iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    return float(np.nanmean(iu))


@pytest.mark.parametrize("name,bb,H,W", [("td4", "resnet18", 33, 65), ("td2", "resnet18", 33, 65), ("td2", "resnet34", 33, 65),
                                         ("td4", "resnet18", 65, 129), ("td2", "resnet18", 49, 81), ("td2", "resnet50", 33, 65),
                                         ("td4", "resnet34", 33, 65), ("td4", "resnet50", 33, 65)])
def test_against_reference_goldens(golden_dir, name, bb, H, W):
    g = np.load(os.path.join(golden_dir, "%s_%s_%dx%d.npz" % (name, bb, H, W)))
    T = 1 + max(int(k.split("_")[0][1:]) for k in g.files if k.startswith("f"))
    spec = arch.model_spec(name, 19, bb)
    h, w = arch.feat_size(H), arch.feat_size(W)
    hk, wk = arch.key_size(h), arch.key_size(w)
    shapes = {"c4": (1, spec.d_model, h, w), "z": (1, spec.d_model, h, w), "v_cur": (1, spec.d_v, h, w), "q_cur": (1, h * w, 64),
              "ln": (1, spec.d_v, h, w), "lowres": (1, 19, h, w), "cache_q": (1, hk * wk, 64), "cache_k": (1, hk * wk, 64),
              "cache_v": (1, hk * wk, spec.d_v)}
    m = make_model(name, bb)
    hist = np.zeros((19, 19), np.int64)
    with torch.no_grad():
        for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
            out = m(torch.from_numpy(x).cuda(), pos_id=t % spec.path_num).cpu().numpy()
            for st, shp in shapes.items():
                key = "f%d_%s" % (t, st)
                if key in g.files:
                    got = m.engine.stage(st, shp)
                    assert np.abs(got - g[key]).max() <= 1e-4 * max(1.0, np.abs(g[key]).max()), (t, st)
            check_frame(out, g["f%d_logits" % t], (name, bb, H, W, t), hist)
    assert clip_miou(hist) >= 0.9995


def _vs_oracle(name, bb, H, W, T, kernel_opts=None):
    spec = arch.model_spec(name, 19, bb)
    ref = tdnet_ref.TDNetRef(spec, weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0))
    m = make_model(name, bb, kernel_opts=kernel_opts)
    tdnet_ref.tune_threads()
    worst, flips = 0.0, 0
    hist = np.zeros((19, 19), np.int64)
    with torch.no_grad():
        for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
            xt = torch.from_numpy(x)
            out = m(xt.cuda(), pos_id=t % spec.path_num).cpu().numpy()
            exp = ref.forward(xt, t % spec.path_num).numpy()
            e, f = check_frame(out, exp, (name, bb, H, W, t), hist)
            worst, flips = max(worst, e), flips + f
    miou = clip_miou(hist)
    print("%s-%s %dx%d %s: worst |dlogit| %.2e, %d label flips over %d frames, clip mIoU %.6f" % (name, bb, H, W, kernel_opts or "", worst, flips, T, miou))
    assert miou >= 0.9995, (name, bb, H, W, miou)


def test_vs_c_operator_oracle():
    """The same gate with the oracle graph running on the plain-C operators (oracle/ops_c.c): no PyTorch kernel on the checking
    side.  Sizes the naive C loops finish in seconds; the warm-up, steady-state and every-path frames of td4 and td2."""
    from oracle import c_ops
    prev = tdnet_ref.set_ops(c_ops.COps)
    try:
        _vs_oracle("td4", "resnet18", 65, 129, 7)
        _vs_oracle("td2", "resnet18", 97, 129, 4)
    finally:
        tdnet_ref.set_ops(prev)


def test_vs_oracle_mid_size():
    _vs_oracle("td4", "resnet18", 257, 513, 11)            # every path in steady state twice
    _vs_oracle("td2", "resnet34", 180, 240, 5)


def test_vs_oracle_c1_512x1024():
    _vs_oracle("td2", "resnet18", 512, 1024, 4)            # BASELINE.json configs[0]


def test_vs_oracle_c2_td2_psp18_1024x2048():
    _vs_oracle("td2", "resnet18", 1024, 2048, 4)           # BASELINE.json configs[1]: warm-up frame + both paths in steady state (d_v = 128 attention at 32768 x 2048)


def test_vs_oracle_td4_resnet50():
    _vs_oracle("td4", "resnet50", 129, 257, 6)             # the third backbone td4_psp18.py:52-66 accepts (d_model = d_v = 2048): attention as four 512-channel launches


def test_vs_oracle_td4_resnet34():
    _vs_oracle("td4", "resnet34", 257, 513, 8)             # td4_psp18.py:52-66 accepts resnet34 as well: every path cold and in steady state


def test_vs_oracle_full_size_td4_1024x2048():
    _vs_oracle("td4", "resnet18", 1024, 2048, 5)           # configs[2]: all four paths cold + first steady-state frame


def test_vs_oracle_td2_psp50_native():
    _vs_oracle("td2", "resnet50", 769, 1537, 3)            # the reference's shipped td2-psp50 at its native resolution (test.py:28-32)


def test_vs_oracle_native_769x1537():
    _vs_oracle("td4", "resnet18", 769, 1537, 5)            # the resolution the reference's LayerNorm([97,193]) fixes


def test_full_size_digest_from_reference(golden_dir):
    """Strided logits samples + statistics captured from the REAL reference at full size (tools/make_golden.py), one digest per
    steady-state frame: at 1024x2048 frames 3..7 are sub-networks 4,1,2,3,4 with a full cache, so every path's attention wiring
    (incl. forward_path3, td4_psp18.py:176-195) is compared with the reference itself, not only with the oracle."""
    g = np.load(os.path.join(golden_dir, "fullsize_digests.npz"))
    # all six tags tools/make_golden.py writes (td2-r18 512x1024 = BASELINE configs[0]'s size, td2-r50 769x1537 = the shipped td2-psp50)
    tags = [("td4", "resnet18", 1024, 2048), ("td2", "resnet18", 1024, 2048), ("td2", "resnet34", 720, 960), ("td4", "resnet18", 769, 1537),
            ("td2", "resnet18", 512, 1024), ("td2", "resnet50", 769, 1537)]
    written = {k[:-len("_last_frame")] for k in g.files if k.endswith("_last_frame") and not k.startswith("psp")}
    assert written == {"%s_%s_%dx%d" % t for t in tags}, written
    for name, bb, H, W in tags:
        tag = "%s_%s_%dx%d" % (name, bb, H, W)
        T = int(g[tag + "_last_frame"]) + 1
        spec = arch.model_spec(name, 19, bb)
        m = make_model(name, bb)
        checked = 0
        with torch.no_grad():
            for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
                out = m(torch.from_numpy(x).cuda(), pos_id=t % spec.path_num)
                if "%s_f%d_sample" % (tag, t) not in g.files:
                    continue
                out = out.cpu().numpy()
                assert np.abs(out[0, :, ::61, ::67] - g["%s_f%d_sample" % (tag, t)]).max() <= 1e-3, (tag, t)
                stats = np.array([out.min(), out.max(), out.mean(), np.sqrt((out.astype(np.float64) ** 2).sum())])
                assert np.allclose(stats, g["%s_f%d_stats" % (tag, t)], rtol=1e-4, atol=1e-4), (tag, t)
                # labels: a sampled label may differ from the reference's only inside the reference's top-2 tie band at that pixel (the
                # digest holds all 19 reference logits of every sampled pixel), and rarely
                ref_s, got_s = g["%s_f%d_sample" % (tag, t)], out[0, :, ::61, ::67]
                bad = got_s.argmax(0) != g["%s_f%d_labels_sample" % (tag, t)]
                assert bad.mean() <= 0.002, (tag, t)
                if bad.any():
                    top2 = np.sort(ref_s, axis=0)[-2:]
                    assert ((top2[1] - top2[0])[bad] <= 2 * max(float(np.abs(got_s - ref_s).max()), 1e-6)).all(), (tag, t, "sampled label flip outside the tie band")
                checked += 1
        assert checked == T - spec.fifo, (tag, checked)


def test_direct_conv_mode_meets_the_same_gate():
    """The library default uses Winograd F(4x4,3x3) for layers 2-4 and the head; the all-direct configuration must stay
    parity-green too."""
    _vs_oracle("td4", "resnet18", 257, 513, 8, kernel_opts={"winograd": 0})
    _vs_oracle("td4", "resnet18", 1024, 2048, 5, kernel_opts={"winograd": 0})


@pytest.mark.parametrize("wino", [0, 3])
def test_uncalibrated_reference_init(wino):
    """Stress with SURVEY 8d's ORIGINAL init (every conv ~ N(0, 2/(k k C_out)), no q/k gain, no depth normalisation): activations
    grow ~8x through the 512->64 projections, scores reach the hundreds, logits the tens -- the absolute 1e-3 gate stops being
    meaningful (the reference's own fp32 CPU evaluation is 1.7e-3 away from an fp64 evaluation of the same graph at this size), so
    the gate is RELATIVE to what fp32 itself can deliver: against an fp64 evaluation of the oracle graph ("truth"),
        max|gpu - truth| <= 4 x max|fp32 CPU oracle - truth|   and   rms(gpu - truth) <= 3 x rms(fp32 CPU oracle - truth),
    and a label may differ from the truth's only inside the truth's top-2 tie band -- asserted exactly as written here (round 2
    asserted a looser 1e-3 max|truth| / 5x rms, which the direct kernel needed: its one sequential fp32 chain over K = 4608 was 6.8x /
    3.5x the CPU's error; the kernel now sums blocks of 512 products, td_conv.h FLUSH).  max|gpu - cpu| is printed as well: it is
    what "within X of the CPU path" means for logits of this magnitude.  Both conv algorithms (direct, Winograd F(4x4); F(2x2) left the
    library in round 5) must pass: Winograd's extra rounding error is a constant factor (measured on the CPU model of the kernels,
    tests/numerics_winograd.py reference-init: rms 1.0x / 2.3x, max 1.8x / 3.0x the direct path's for F(2x2) / F(4x4)), not something the
    calibrated weights were hiding.

    THREE clips, pooled (round 5).  Both errors are heavy-tailed draws, not constants of the kernels: with these weights the attention is
    nearly one-hot, and which queries sit near a tie decides the error.  Measured on one box (profiles/r05k_*): over clip seeds 1-3, either
    7x7 stem kernel and both conv algorithms max|gpu - truth| is 2.7e-3 .. 1.06e-2 and the CPU path's OWN max error 1.1e-3 .. 8.0e-3 -- it
    also moves between two calls on the same clip (3.3e-3 / 8.0e-3: oneDNN's thread partition).  Clip 1 alone is the CPU's luckiest draw
    (1.1e-3); re-ordering the stem's 147 products (the packed-row image, fusion bit 65536; the operator alone is 9.33e-8 rms from fp64 with
    either kernel, tools/stem_numerics.py) moved the GPU's draw on it from 2.4x to 7.6x while clips 2 and 3 read 0.55x and 1.5x.  So the
    same 4x / 3x gate is asserted on the statistics pooled over the three clips (measured 1.1x / 1.9x Winograd, 1.1x / 1.2x direct), and every
    clip on its own against max|gpu - truth| <= 5e-4 max|truth| (measured <= 2.6e-4)."""
    runs = [_reference_init_stress(129, 257, 5, wino, seed=seed, gate=False) for seed in (1, 2, 3)]
    e_gpu, e_cpu = max(r["e_gpu"] for r in runs), max(r["e_cpu"] for r in runs)
    n = sum(r["n"] for r in runs)
    r_gpu, r_cpu = (sum(r["s_gpu"] for r in runs) / n) ** 0.5, (sum(r["s_cpu"] for r in runs) / n) ** 0.5
    print("pooled over 3 clips, winograd=%d: max err gpu %.2e vs cpu-fp32 %.2e (x%.2f); rms gpu %.2e vs cpu-fp32 %.2e (x%.2f)"
          % (wino, e_gpu, e_cpu, e_gpu / e_cpu, r_gpu, r_cpu, r_gpu / r_cpu))
    assert e_gpu <= 4.0 * e_cpu and r_gpu <= 3.0 * r_cpu, (wino, e_gpu, e_cpu, r_gpu, r_cpu)
    for r in runs:
        assert r["e_gpu"] <= 5e-4 * r["tmax"], (wino, r)


def _reference_init_stress(H, W, T, wino, gc_bound=None, extra_opts=None, seed=1, gate=True):
    spec = arch.model_spec("td4", 19, "resnet18")
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0, init="reference")
    ref32 = tdnet_ref.TDNetRef(spec, sd)
    ref64 = tdnet_ref.TDNetRef(spec, {k: torch.from_numpy(np.asarray(v)).double() for k, v in sd.items()})
    m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, kernel_opts=dict({"winograd": wino}, **(extra_opts or {}))).eval().to("cuda")
    m.load_state_dict(sd)
    tdnet_ref.tune_threads()
    e_gpu, e_cpu, s_gpu, s_cpu, n, tmax, e_gc = 0.0, 0.0, 0.0, 0.0, 0, 0.0, 0.0
    edges = np.array([0.0, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, np.inf])
    gap_hist, flip_hist, flips_gc = np.zeros(6, np.int64), np.zeros(6, np.int64), 0
    with torch.no_grad():
        for t, x in enumerate(weights.synth_video(H, W, T, seed=seed)):
            xt = torch.from_numpy(x)
            out = m(xt.cuda(), pos_id=t % 4).cpu().double().numpy()
            truth = ref64.forward(xt.double(), t % 4).numpy()
            cpu = ref32.forward(xt, t % 4).double().numpy()
            dg, dc = out - truth, cpu - truth
            e_gc = max(e_gc, float(np.abs(out - cpu).max()))
            tmax = max(tmax, float(np.abs(truth).max()))
            e_gpu, e_cpu = max(e_gpu, float(np.abs(dg).max())), max(e_cpu, float(np.abs(dc).max()))
            s_gpu, s_cpu, n = s_gpu + float((dg ** 2).sum()), s_cpu + float((dc ** 2).sum()), n + dg.size
            top2 = np.sort(truth[0], axis=0)[-2:]
            gap = top2[1] - top2[0]
            bad = out[0].argmax(0) != truth[0].argmax(0)
            gap_hist += np.histogram(gap, edges)[0]
            flip_hist += np.histogram(gap[bad], edges)[0]
            flips_gc += int((out[0].argmax(0) != cpu[0].argmax(0)).sum())
            if bad.any():
                assert (gap[bad] <= 2 * float(np.abs(dg).max())).all(), (wino, t, "label flip outside the tie band")
    r_gpu, r_cpu = (s_gpu / n) ** 0.5, (s_cpu / n) ** 0.5
    print("reference init %dx%d, winograd=%d, %d frames: max|truth| %.1f; max err gpu %.2e (%.1e of max|truth|) vs cpu-fp32 %.2e (x%.2f); rms gpu %.2e vs cpu-fp32 %.2e (x%.2f); max|gpu - cpu| %.2e (x%.2f of the cpu's own error)"
          % (H, W, wino, T, tmax, e_gpu, e_gpu / tmax, e_cpu, e_gpu / e_cpu, r_gpu, r_cpu, r_gpu / r_cpu, e_gc, e_gc / e_cpu))
    print("  top-2 gap of the fp64 truth, pixels per decade [0,1e-4) [1e-4,1e-3) [1e-3,1e-2) [1e-2,1e-1) [1e-1,1) [1,inf): %s; labels differing from the truth per decade: %s; labels differing from the fp32 CPU path: %d of %d"
          % (gap_hist.tolist(), flip_hist.tolist(), flips_gc, int(gap_hist.sum())))
    assert not gate or (e_gpu <= 4.0 * e_cpu and r_gpu <= 3.0 * r_cpu), (wino, tmax, e_gpu, e_cpu, r_gpu, r_cpu)
    if gc_bound is not None:
        assert e_gc <= gc_bound * e_cpu, (e_gc, e_cpu)
    return {"e_gpu": e_gpu, "e_cpu": e_cpu, "e_gc": e_gc, "s_gpu": s_gpu, "s_cpu": s_cpu, "n": n, "tmax": tmax}


def test_reference_init_at_the_checkpoints_geometry_769x1537():
    """The same stress where a real checkpoint lives: td4-psp18 at 769x1537 with the [97,193] LayerNorm affine
    (td4_psp18.py:107-110), SURVEY 8d's un-calibrated init (resnet.py:162-165), P + 2 frames, the default conv algorithm (Winograd
    F(4x4)).  Same relative gate (4x max / 3x rms of the fp32 CPU path's own distance to an fp64 evaluation, flips only in the tie band),
    plus the distance to the CPU path itself: max|gpu - cpu| <= 3 x max|cpu - truth| (the triangle inequality alone allows 5 x; 2.0 x
    was measured at 129x257).  The printed histogram says how many pixels sit in which decade of the truth's top-2 gap and how many of
    them the GPU labels differently: "labels differ only inside the tie band", quantified at real logit magnitudes."""
    _reference_init_stress(769, 1537, 6, 3, gc_bound=3.0)


def test_two_models_with_different_kernel_options_in_one_process():
    """Per-handle options: an all-direct td2 and a default (Winograd F(4x4), single-pass attention) td2 side by side, fed alternately,
    each reproduces what it computes alone, and they differ from each other only by rounding."""
    H, W = 129, 257
    frames = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, 4, seed=2)]
    with torch.no_grad():
        alone = {}
        for key, opts in (("direct", {"winograd": 0, "attention": 0, "fusion": 0}), ("default", None)):
            m = make_model("td2", "resnet18", kernel_opts=opts)
            alone[key] = [m(x, pos_id=t % 2).clone() for t, x in enumerate(frames)]
        a, b = make_model("td2", "resnet18", kernel_opts={"winograd": 0, "attention": 0, "fusion": 0}), make_model("td2", "resnet18")
        assert a.engine is None and b.engine is None
        for t, x in enumerate(frames):
            oa, ob = a(x, pos_id=t % 2), b(x, pos_id=t % 2)
            assert torch.equal(oa, alone["direct"][t]) and torch.equal(ob, alone["default"][t])
            assert 0 < (oa - ob).abs().max().item() < 1e-3
        assert a.engine.opts()["winograd"] == 0 and b.engine.opts()["winograd"] == 3


def test_properties_determinism_labels_reset():
    H, W = 129, 257
    frames = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, 6, seed=3)]
    m = make_model("td4", "resnet18", seed=5)
    with torch.no_grad():
        a = [m(x, pos_id=t % 4).clone() for t, x in enumerate(frames)]
        assert m.engine.fifo_len() == 3
        m.reset()
        assert m.engine.fifo_len() == 0
        b = [m(x, pos_id=t % 4).clone() for t, x in enumerate(frames)]
        for x, y in zip(a, b):
            assert torch.equal(x, y)                               # bit-identical replay after reset
        m.reset()
        for t, x in enumerate(frames):
            lab = m.forward_labels(x, pos_id=t % 4)
            assert torch.equal(lab[0].long(), a[t][0].max(0)[1])   # == output.max(1)[1] (test.py:61), first max wins
        # warm-up frames ignore the cache: frame 0 of a fresh stream does not depend on what came before reset
        m.reset()
        assert torch.equal(m(frames[0], pos_id=0), a[0])
    # wrong-resolution weights fail like the reference's LayerNorm([97,193]) (td4_psp18.py:107-110)
    spec = arch.model_spec("td4", 19, "resnet18")
    m2 = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None).eval()
    m2.load_state_dict(weights.synth_state_dict(spec, 97, 193, 0))
    with pytest.raises(RuntimeError):
        m2(frames[0], pos_id=0)


def test_every_schedule_option_reproduces_the_default_bit_for_bit():
    """tdnet_opts.overlap / fusion choose WHEN and on which kernel variant the same products are summed in the same order: no chains
    (round 2's schedule), chains with 1 / 4 channels per lane, the persistent or the LDS-DMA-fed Winograd GEMM; in the fp16 mode the
    tap-by-tap or the row-image conv kernel, the conv without / with dedicated loader waves (the default since round 4), wide instead of narrow
    tiles (layer1 on the register-staged kernel instead of the narrow LDS-DMA tiles: round 5), one launch per register-staged conv instead of
    the grouped launches (fusion bit 131072, default since round 5: every fp16 variant below lacks it).  On the real streams and DMA engines (the
    emulator runs them in issue order) every variant must give the default's logits bit for bit, frame by frame, through warm-up and steady
    state, including a repeated pos_id.  (The schedules that lost in rounds 3-4 were removed in round 5 and left this matrix.)"""
    H, W = 257, 513
    frames = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, 8, seed=11)]
    pos = [0, 1, 2, 3, 0, 0, 1, 2]
    with torch.no_grad():
        for base, variants in (({}, [{"overlap": 0}, {"overlap": 1 | 4}, {"overlap": 33 | 4}, {"overlap": 41 | 4}, {"overlap": 8}]),   # bit 4: the chains on this small map too
                               ({"precision": 1}, [{"fusion": 6 | 2048}, {"fusion": 6 | 1024}, {"fusion": 38}, {"fusion": 38 | 8192}])):
            m = make_model("td4", "resnet18", seed=3, kernel_opts=dict(base))
            ref = [m(x, pos_id=p).clone() for x, p in zip(frames, pos)]
            m.engine.close()
            for v in variants:
                mv = make_model("td4", "resnet18", seed=3, kernel_opts=dict(base, **v))
                for t, (x, p) in enumerate(zip(frames, pos)):
                    out = mv(x, pos_id=p)
                    assert torch.equal(out, ref[t]), (base, v, t, float((out - ref[t]).abs().max()))
                for k, val in v.items():
                    assert mv.engine.opts()[k] == val
                mv.engine.close()


def test_batch_of_streams_and_non_strict_loading():
    """nn.Module contract edges.  (1) A batch of N frames is N independent streams in the reference (queue entries
    are [N, Lk, .], every plane LayerNorm / softmax is per sample: td4_psp18.py:123-154,216-229): sample i of model(batch) must be bit
    for bit what a model of its own computes on stream i, through warm-up and steady state; the batch size cannot change while frames
    are cached.  (2) load_state_dict(strict=False) drops unexpected keys, reports missing ones, and the missing tensors come from the
    seeded generator (without a seed the first frame raises instead of running on unspecified values)."""
    H, W, T = 65, 129, 6
    a = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, T, seed=21)]
    b = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, T, seed=22)]
    with torch.no_grad():
        ma, mb, m2 = make_model("td4", "resnet18", seed=4), make_model("td4", "resnet18", seed=4), make_model("td4", "resnet18", seed=4)
        refs = []
        for t in range(T):
            ra, rb = ma(a[t], pos_id=t % 4), mb(b[t], pos_id=t % 4)
            out = m2(torch.cat([a[t], b[t]], 0), pos_id=t % 4)
            assert out.shape == (2, 19, H, W)
            assert torch.equal(out[0:1], ra) and torch.equal(out[1:2], rb), t
            refs.append(torch.cat([ra, rb], 0))
        # sample 1 runs on a stream of its own beside the caller's: a caller on a NON-default stream must find both samples complete
        # when its own stream gets there (the join), without a device synchronisation in between
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            m5 = make_model("td4", "resnet18", seed=4)
            sums = [m5(torch.cat([a[t], b[t]], 0), pos_id=t % 4).double().sum(dim=(1, 2, 3)) for t in range(T)]
        side.synchronize()
        for t in range(T):
            assert torch.equal(sums[t], refs[t].double().sum(dim=(1, 2, 3))), t
        # three samples on the two lanes: samples 0 and 2 share the caller's stream, sample 1 runs beside them
        c = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, T, seed=23)]
        mc, m6 = make_model("td4", "resnet18", seed=4), make_model("td4", "resnet18", seed=4)
        for t in range(T):
            out = m6(torch.cat([a[t], b[t], c[t]], 0), pos_id=t % 4)
            assert torch.equal(out[0:2], refs[t]) and torch.equal(out[2:3], mc(c[t], pos_id=t % 4)), t
        # the samples of a batch share ONE weight block (include/tdnet.h tdnet_create_shared; td4_psp18.py:216-229: one module, N samples):
        # three handles on m6's block, each costing only its workspace + FIFO; the block outlives the handle that loaded it
        wb, hb, refs_n = m6.engine.memory_bytes()
        assert refs_n == 3 and wb > 50e6 and all(e.memory_bytes() == (wb, hb, 3) for e in m6._extra_engines)
        survivor = m6._extra_engines.pop()
        m6._close_engines()                                             # the owner and one sharer go first
        assert survivor.memory_bytes() == (wb, hb, 1)
        o1 = torch.empty((1, 19, H, W), device="cuda")
        survivor.reset()
        survivor.warmup(torch.cuda.current_stream().cuda_stream)       # explicit placement of its internal streams (host-synchronising, once)
        survivor.forward(a[0].data_ptr(), 0, o1.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert torch.equal(o1, make_model("td4", "resnet18", seed=4)(a[0], pos_id=0))
        survivor.close()
        lab = m2.forward_labels(torch.cat([a[0], b[0]], 0), pos_id=(T % 4))
        assert lab.shape == (2, H, W)
        with pytest.raises(RuntimeError, match="batch size"):
            m2(a[0], pos_id=0)                                          # frames of a 2-stream batch are cached
        m2.reset()
        assert torch.equal(m2(a[0], pos_id=0), make_model("td4", "resnet18", seed=4)(a[0], pos_id=0))
        # strict=False
        spec = arch.model_spec("td4", 19, "resnet18")
        sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 4)
        part = {k: v for k, v in sd.items() if not k.startswith("head3.")}
        part["module.extra"] = np.zeros(3, np.float32)
        m3 = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, synthetic_seed=4).eval().to("cuda")
        res = m3.load_state_dict(part, strict=False)
        assert res.unexpected_keys == ["module.extra"] and res.missing_keys and all(k.startswith("head3.") for k in res.missing_keys)
        assert torch.equal(m3(a[0], pos_id=0), make_model("td4", "resnet18", seed=4)(a[0], pos_id=0))   # the seeded generator filled head3.*
        m4 = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None).eval().to("cuda")
        m4.load_state_dict(part, strict=False)
        with pytest.raises(RuntimeError, match="strict=False"):
            m4(a[0], pos_id=0)
        with pytest.raises(RuntimeError):
            m4.load_state_dict(part, strict=True); m4(a[0], pos_id=0)   # strict: the unexpected / missing keys are errors (td4_psp18.py:237)
