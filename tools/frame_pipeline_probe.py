#!/usr/bin/env python3
"""GPU-box probe: ONE clip, two frames in flight (parallel.FramePipelinedStream: two handles on two HIP streams, the frames' cache entries
handed over between encode and propagate) against the same clip on one handle.

    python tools/frame_pipeline_probe.py [--model td2] [--backbone resnet34] [--size 720x960] [--precision fp16] [--rounds 60]

Prints frames/s of both and checks the pipelined outputs bit for bit against the single handle's."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="td2")
    ap.add_argument("--backbone", default="resnet34")
    ap.add_argument("--size", default="720x960")
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--lane-opts", default="", help="tdnet_opts of the two pipelined handles only, k=v,... (the single handle keeps the defaults)")
    a = ap.parse_args()
    import torch
    from tdnet_amd import arch, parallel, weights
    from tdnet_amd.model import td2_psp50, td4_psp18
    H, W = (int(v) for v in a.size.lower().split("x"))
    spec = arch.model_spec(a.model, 19, a.backbone)
    P = spec.path_num
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    dev = torch.device("cuda", 0)
    NF = 8
    clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100)]
    cls = td4_psp18.td4_psp18 if a.model == "td4" else td2_psp50.td2_psp50
    opts = {"precision": 1} if a.precision == "fp16" else {}

    lane_opts = {k.strip(): int(v) for k, _, v in (p.partition("=") for p in a.lane_opts.split(",") if p.strip())}

    def make(extra=None):
        m = cls(nclass=19, path_num=P, model_path=None, backbone=a.backbone, kernel_opts=dict(opts, **(extra or {}))).eval().to(dev)
        m.load_state_dict(sd)
        return m
    n = 2 * a.rounds
    with torch.no_grad():
        one = make()
        ref = [one(clip[t % NF], pos_id=t % P).clone() for t in range(P + 5)]
        t = P + 5
        for _ in range(8):
            one(clip[t % NF], pos_id=t % P); t += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            one(clip[t % NF], pos_id=t % P); t += 1
        host_one = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        fps_one = n / (time.perf_counter() - t0)
        one.engine.close()
        del one
        stages = [make(lane_opts), make(lane_opts)]
        fp = parallel.FramePipelinedStream(stages, P, dev, (H, W))
        outs = fp.process([clip[t % NF] for t in range(P + 5)], first_frame=0)          # odd count: a short last round
        same = all(torch.equal(x, y) for x, y in zip(outs, ref))
        del outs
        t = P + 5
        fp.process([clip[t % NF]], first_frame=t); t += 1                             # back to an even frame number
        for _ in range(4):
            fp.process([clip[t % NF], clip[(t + 1) % NF]], first_frame=t, join=False); t += 2
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.rounds):
            fp.process([clip[t % NF], clip[(t + 1) % NF]], first_frame=t, join=False); t += 2
        host_two = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        fps_two = n / (time.perf_counter() - t0)
    print("%s-psp%s %dx%d %s: one handle %.1f frames/s (host loop %.0f us/frame); two frames in flight %.1f frames/s (x %.3f; host loop %.0f us/frame); outputs %s" %
          (a.model, a.backbone[6:], H, W, a.precision, fps_one, host_one, fps_two, fps_two / fps_one, host_two,
           ("bit-identical to the single handle" if same else "DIFFER from the single handle") + ("" if not lane_opts else "  [lanes: %s]" % a.lane_opts)))


if __name__ == "__main__":
    main()
