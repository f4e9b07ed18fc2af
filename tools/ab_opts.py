#!/usr/bin/env python3
"""Same-process A/B of tdnet_opts variants on one workload (GPU box).

    python tools/ab_opts.py [--model td4] [--backbone resnet18] [--size 1024x2048] [--precision fp32|fp16] [--steps 60] [--rounds 3] \
        "" "overlap=0" "overlap=33" "precision=1,overlap=0"

One process, one set of synthetic weights and frames; per round every variant gets a fresh handle (an idle handle's streams slow a busy
one: INTEGRATION.md 3), P + 4 warm-up frames and `steps` timed frames between two device synchronisations; the rounds interleave the
variants (A B C A B C ...) so that clock / thermal drift of the box spreads over all of them.  Also reports whether each variant's logits
of a fixed 6-frame replay are bit-identical to the first variant's.  Costs ~3 s per variant and round instead of a bench.py process each.

Frame budget by leaving pieces out (DESIGN_experiments 8.6): a NON-shipping library built with TDNET_EXTRA_CXXFLAGS=-DTDNET_TIMING_PROBES (the
variable must be set both where the library is built and where it is loaded, it is part of the build stamp) honours TDNET_PROBE_SKIP=<mask>
(1 no Winograd transforms, 2 no cache-only chain, 4 no final attention, 8 no Winograd GEMMs; results are garbage, timing only):
    export TDNET_EXTRA_CXXFLAGS=-DTDNET_TIMING_PROBES; for m in 0 1 2 4 8; do TDNET_PROBE_SKIP=$m python tools/ab_opts.py --rounds 2 ""; done"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_variant(text):
    out = {}
    for part in text.split(","):
        part = part.strip()
        if part:
            k, _, v = part.partition("=")
            out[k.strip()] = int(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="td4")
    ap.add_argument("--backbone", default="resnet18")
    ap.add_argument("--size", default="1024x2048")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--sync-each", action="store_true", help="synchronise the device after every frame (latency mode: the host never runs ahead)")
    ap.add_argument("--json", default=None, help="also write the table as JSON lines to this file")
    ap.add_argument("variants", nargs="*", default=[""])
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
    clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100)]
    cls = td4_psp18.td4_psp18 if a.model == "td4" else td2_psp50.td2_psp50
    base = {"precision": 1} if a.precision == "fp16" else {}
    variants = [parse_variant(v) for v in a.variants]
    fps = [[] for _ in variants]
    host_us = [[] for _ in variants]
    launches = [0] * len(variants)
    ident = [None] * len(variants)
    ref_out = None
    for r in range(a.rounds):
        for i, v in enumerate(variants):
            m = cls(nclass=19, path_num=P, model_path=None, backbone=a.backbone, kernel_opts=dict(base, **v)).eval().to(dev)
            m.load_state_dict(sd)
            with torch.no_grad():
                if r == 0:                                             # fixed replay: frames 0..P+1 from an empty FIFO
                    outs = [m(clip[t % NF], pos_id=t % P).clone() for t in range(P + 2)]
                    if ref_out is None:
                        ref_out = outs
                        ident[i] = "reference"
                    else:
                        d = max(float((x - y).abs().max()) for x, y in zip(outs, ref_out))
                        ident[i] = "bit-identical" if all(torch.equal(x, y) for x, y in zip(outs, ref_out)) else "max|d| %.2e" % d
                    del outs
                t = P + 2 if r == 0 else 0
                for _ in range(P + 4):
                    m(clip[t % NF], pos_id=t % P); t += 1
                torch.cuda.synchronize(dev)
                th = time.perf_counter()
                for _ in range(2):                                     # host cost of enqueueing a frame on an empty queue
                    m(clip[t % NF], pos_id=t % P); t += 1
                host_us[i].append((time.perf_counter() - th) / 2 * 1e6)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    m(clip[t % NF], pos_id=t % P); t += 1
                    if a.sync_each:
                        torch.cuda.synchronize(dev)
                torch.cuda.synchronize(dev)
                fps[i].append(a.steps / (time.perf_counter() - t0))
                launches[i] = m.engine.last_launch_count()
            m.engine.close()
            del m
    print("%s-psp%s %dx%d %s, %d steps x %d rounds (interleaved), frames/s:" % (a.model, a.backbone[6:], H, W, a.precision + (", device synchronised after every frame" if a.sync_each else ""), a.steps, a.rounds))
    rows = []
    for v, f, idn, hu, nl in zip(a.variants, fps, ident, host_us, launches):
        med = statistics.median(f)
        rows.append({"variant": v or "(default)", "fps_median": round(med, 2), "fps_rounds": [round(x, 2) for x in f], "vs_first": idn,
                     "host_enqueue_us_per_frame": round(statistics.median(hu), 1), "sync_each_frame": bool(a.sync_each), "launches_per_frame": nl})
        print("  %-44s median %8.2f   rounds %s   host enqueue %6.0f us/frame   %3d launches/frame   %s" % (v or "(default)", med, " ".join("%.1f" % x for x in f), statistics.median(hu), nl, idn))
    if a.json:
        with open(a.json, "a") as f:
            for row in rows:
                f.write(json.dumps(dict(row, workload="%s-psp%s %dx%d %s" % (a.model, a.backbone[6:], H, W, a.precision))) + "\n")


if __name__ == "__main__":
    main()
