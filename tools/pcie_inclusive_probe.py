#!/usr/bin/env python3
"""GPU-box probe: frames/s when every frame STARTS IN HOST MEMORY and its labels END there (the reference harness' `image.to(device)` ...
`output.max(1)[1].cpu()`, Testing/test.py:47-61), beside bench.py's rate with the clip resident in HBM.

  resident   bench.py's loop: frames pre-staged in HBM, logits stay there
  literal    the harness' loop: pageable host tensor -> .to(device) -> model() -> .max(1)[1].cpu(), one frame at a time
  pipelined  pinned host buffers, H2D of frame t + 1 on a copy stream under the compute of frame t, forward_labels (int32 labels, the
             full-resolution logits are never written) and an asynchronous D2H of the labels

    python tools/pcie_inclusive_probe.py [--model td4] [--backbone resnet18] [--size 1024x2048] [--steps 60]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="td4")
    ap.add_argument("--backbone", default="resnet18")
    ap.add_argument("--size", default="1024x2048")
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    import torch
    from tdnet_amd import arch, weights
    from tdnet_amd.model import td2_psp50, td4_psp18
    H, W = (int(v) for v in a.size.lower().split("x"))
    spec = arch.model_spec(a.model, 19, a.backbone)
    P = spec.path_num
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    dev = torch.device("cuda", 0)
    NF = 8
    host = [torch.from_numpy(x) for x in weights.synth_video(H, W, NF, seed=100)]
    cls = td4_psp18.td4_psp18 if a.model == "td4" else td2_psp50.td2_psp50
    m = cls(nclass=19, path_num=P, model_path=None, backbone=a.backbone).eval().to(dev)
    m.load_state_dict(sd)
    print("%s-psp%s %dx%d fp32: a frame is %.1f MB host -> device, labels %.1f MB (int64, literal) / %.1f MB (int32, pipelined) device -> host" %
          (a.model, a.backbone[6:], H, W, 3 * H * W * 4 / 1e6, H * W * 8 / 1e6, H * W * 4 / 1e6))
    with torch.no_grad():
        # resident
        res = [x.to(dev) for x in host]
        t = 0
        for _ in range(P + 4):
            m(res[t % NF], pos_id=t % P); t += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m(res[t % NF], pos_id=t % P); t += 1
        torch.cuda.synchronize()
        print("  resident  %8.1f frames/s" % (a.steps / (time.perf_counter() - t0)))
        # literal harness loop
        labels_lit = None
        for _ in range(4):
            labels_lit = m(host[t % NF].to(dev), pos_id=t % P).max(1)[1].cpu(); t += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            labels_lit = m(host[t % NF].to(dev), pos_id=t % P).max(1)[1].cpu(); t += 1
        torch.cuda.synchronize()
        print("  literal   %8.1f frames/s" % (a.steps / (time.perf_counter() - t0)))
        # pipelined
        pin_in = [x.pin_memory() for x in host]
        dev_in = [torch.empty_like(res[0]) for _ in range(2)]
        pin_out = [torch.empty((1, H, W), dtype=torch.int32).pin_memory() for _ in range(2)]
        copy_s, out_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        in_ready = [torch.cuda.Event() for _ in range(2)]
        in_free = [torch.cuda.Event() for _ in range(2)]
        out_done = [torch.cuda.Event() for _ in range(2)]
        for e in in_free + out_done:
            e.record(cur)

        def upload(frame, slot):
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(in_free[slot])
                dev_in[slot].copy_(pin_in[frame % NF], non_blocking=True)
                in_ready[slot].record(copy_s)

        def run(nsteps, t):
            upload(t, t & 1)
            for _ in range(nsteps):
                slot = t & 1
                upload(t + 1, slot ^ 1)
                cur.wait_event(in_ready[slot])
                lab = m.forward_labels(dev_in[slot], pos_id=t % P)
                in_free[slot].record(cur)
                done = torch.cuda.Event()
                done.record(cur)
                with torch.cuda.stream(out_s):
                    out_s.wait_event(done)
                    out_done[slot].synchronize()                       # the host has consumed this slot's previous labels (here: nothing to do)
                    pin_out[slot].copy_(lab, non_blocking=True)
                    lab.record_stream(out_s)
                    out_done[slot].record(out_s)
                t += 1
            return t

        t = run(6, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = run(a.steps, t)
        torch.cuda.synchronize()
        print("  pipelined %8.1f frames/s" % (a.steps / (time.perf_counter() - t0)))
    m._close_engines()


if __name__ == "__main__":
    main()
