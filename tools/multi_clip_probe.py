#!/usr/bin/env python3
"""GPU-box probe: K independent clips on ONE GPU, one handle and one HIP stream per clip, frames enqueued round-robin from one host thread.

    python tools/multi_clip_probe.py [--model td2] [--backbone resnet34] [--size 720x960] [--precision fp16] [--steps 120] [--clips 1,2,3,4]

A frame of a small map leaves CUs idle (720x960 fp16: most launches are < 256 workgroups and a K loop is a latency chain), so frames of
DIFFERENT clips can run beside each other.  Prints the aggregate frames/s per K and checks that every clip's logits equal the ones the
same clip produces alone (bit for bit: the handles share nothing but the weights' values)."""
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
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--clips", default="1,2,3,4")
    ap.add_argument("--offset-us", type=float, default=0.0, help="delay the odd clips' streams once by this much before the timed loop (phase offset between the lanes)")
    ap.add_argument("--batched", action="store_true", help="ONE model fed [K, 3, H, W] batches (model/_base.py: sample i on its own handle and stream) "
                                                          "instead of K models on K caller streams")
    a = ap.parse_args()
    import torch
    import tdnet_amd
    from tdnet_amd import arch, weights
    from tdnet_amd.model import td2_psp50, td4_psp18
    H, W = (int(v) for v in a.size.lower().split("x"))
    spec = arch.model_spec(a.model, 19, a.backbone)
    P = spec.path_num
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    dev = torch.device("cuda", 0)
    NF = 6
    cls = td4_psp18.td4_psp18 if a.model == "td4" else td2_psp50.td2_psp50
    opts = {"precision": 1} if a.precision == "fp16" else {}
    ks = [int(v) for v in a.clips.split(",")]
    kmax = max(ks)
    clips = [[torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100 + c)] for c in range(kmax)]
    print("%s-psp%s %dx%d %s, GPU_MAX_HW_QUEUES=%s" % (a.model, a.backbone[6:], H, W, a.precision, os.environ.get("GPU_MAX_HW_QUEUES")))
    alone = {}
    with torch.no_grad():                                              # every clip alone, on the default stream
        m = cls(nclass=19, path_num=P, model_path=None, backbone=a.backbone, kernel_opts=dict(opts)).eval().to(dev)
        m.load_state_dict(sd)
        for c in range(kmax):
            m.reset()
            alone[c] = [m(clips[c][t % NF], pos_id=t % P).clone() for t in range(P + 2)]
        torch.cuda.synchronize(dev)
        m.engine.close()
        del m
    for K in ks if not a.batched else []:
        models = []
        for c in range(K):
            m = cls(nclass=19, path_num=P, model_path=None, backbone=a.backbone, kernel_opts=dict(opts)).eval().to(dev)
            m.load_state_dict(sd)
            models.append(m)
        # which hardware queue a new stream lands on depends on how many streams the process has created (a lottery: the same K = 2 ran
        # 1059 or 1445 frames/s): every stream is placed beside the ones taken so far with the library's spin-pair test
        from tdnet_amd import _capi
        from tdnet_amd.model._base import _TDNetBase
        streams = []
        for _ in range(K):
            streams.append(_TDNetBase._stream_beside(streams, dev, _capi.lib()) if streams else torch.cuda.Stream(dev))
        with torch.no_grad():
            outs = [[] for _ in range(K)]
            for t in range(P + 2):                                     # replay from an empty FIFO: compared with the clip alone
                for c in range(K):
                    with torch.cuda.stream(streams[c]):
                        outs[c].append(models[c](clips[c][t % NF], pos_id=t % P).clone())
            torch.cuda.synchronize(dev)
            same = all(torch.equal(x, y) for c in range(K) for x, y in zip(outs[c], alone[c]))
            del outs
            t = P + 2
            for _ in range(P + 4):
                for c in range(K):
                    with torch.cuda.stream(streams[c]):
                        models[c](clips[c][t % NF], pos_id=t % P)
                t += 1
            torch.cuda.synchronize(dev)
            if a.offset_us > 0 and K > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize(dev)
                per_us = 1000000 / (e0.elapsed_time(e1) * 1e3)          # _sleep cycles per microsecond
                for c in range(1, K, 2):
                    with torch.cuda.stream(streams[c]):
                        torch.cuda._sleep(int(a.offset_us * per_us))
            t0 = time.perf_counter()
            for _ in range(a.steps):
                for c in range(K):
                    with torch.cuda.stream(streams[c]):
                        models[c](clips[c][t % NF], pos_id=t % P)
                t += 1
            th = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        print("  %d clip(s): %8.1f frames/s aggregate (%.1f per clip), host enqueue %.0f us per frame, logits %s" %
              (K, K * a.steps / dt, a.steps / dt, th / (K * a.steps) * 1e6, "bit-identical to the clips alone" if same else "DIFFER from the clips alone"))
        for m in models:
            m.engine.close()
        del models, streams
    for K in ks if a.batched else []:
        with torch.no_grad():
            m = cls(nclass=19, path_num=P, model_path=None, backbone=a.backbone, kernel_opts=dict(opts)).eval().to(dev)
            m.load_state_dict(sd)
            frames = [torch.cat([clips[c][t] for c in range(K)], 0) for t in range(NF)]
            same = True
            for t in range(P + 2):
                out = m(frames[t % NF], pos_id=t % P)
                same = same and all(torch.equal(out[c:c + 1], alone[c][t]) for c in range(K))
            t = P + 2
            for _ in range(P + 4):
                m(frames[t % NF], pos_id=t % P); t += 1
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                m(frames[t % NF], pos_id=t % P); t += 1
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            print("  batch of %d: %8.1f frames/s aggregate (%.1f batches/s), logits %s" %
                  (K, K * a.steps / dt, a.steps / dt, "bit-identical to the clips alone" if same else "DIFFER from the clips alone"))
            m._close_engines()
            del m


if __name__ == "__main__":
    main()
