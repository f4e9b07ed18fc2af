#!/usr/bin/env python3
"""bench.py -- frames/s of the TDNet per-frame hot path on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model td4|td2] [--size HxW]

A "step" is one frame of one synthetic video clip through model(image, pos_id) (Testing/test.py:53): sub-network
forward + attention propagation from the cached frames + head + x8 upsample to full-resolution logits.  Frames are
pre-staged in HBM (the timed region holds no H2D copy); every rank serves its own clip (weak scaling, no per-frame
collective); weights are generated on rank 0 and broadcast once over RCCL.  Timing: barrier + synchronize, EXACTLY K
steps, synchronize + barrier, max over ranks.

Extra objects on the JSON line:
  roofline      the dominant kernel (128x128-tile 3x3 fp32-MFMA implicit-GEMM conv): algorithmic FLOP per launch /
                average launch duration, measured with HIP events on the forward's stream during a profiled replay of
                the same steps right after the timed region; peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (oracle/tdnet_ref.py, the reference's op graph on torch-CPU/oneDNN, all host cores)
                timed on a bounded sample of the same clip: steady-state frames after the P warm-up frames.
  parity        GPU logits vs that oracle on the sampled frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tdnet_amd import arch, parallel, weights  # noqa: E402
from tdnet_amd.model import pspnet, td2_psp50, td4_psp18  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3


def traffic_from_profiles():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the
    gfx950 calibration + WRITE_SIZE; counters cannot be read live inside bench.py).  None if no profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")) as f:
            return round(json.load(f)["traffic_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--model", default="td4", choices=["td4", "td2", "psp"])
    ap.add_argument("--backbone", default=None, help="resnet18 (default) | resnet34 | resnet50 (td2) | resnet101 (psp)")
    ap.add_argument("--size", default="1024x2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-pipeline", type=int, default=None, help="tuning: 0 one-stage / 1 two-stage conv prefetch (default: library default)")
    ap.add_argument("--cpu-frames", type=int, default=2, help="steady-state frames timed on the CPU oracle")
    ap.add_argument("--winograd", type=int, default=None, help="conv algorithm: 0 direct, 1 Winograd F(2x2,3x3) for layers 3-4 + head, 3 Winograd F(4x4,3x3) for them (default: library default)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"],
                    help="fp32 (default; the mode the parity gate is defined for) | fp16 = fp16-input MFMA convs, fp32 accumulate "
                         "(BASELINE config 5); parity vs the fp32 CPU path is reported, not gated")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="independent clips served concurrently by one GPU, each with its own handle/FIFO on its own HIP stream "
                         "(throughput mode; a step is then one frame of EVERY clip).  Default 1 = BASELINE's one clip per GPU")
    ap.add_argument("--mode", default="clips", choices=["clips", "path-parallel"],
                    help="clips (default, BASELINE): independent clips, one per GPU, no per-frame communication | path-parallel: ONE "
                         "stream served by all N ranks, rank g takes frames t = g (mod N), one all-gather of cache entries per round "
                         "of N frames (a step is then one round)")
    args = ap.parse_args()
    H, W = (int(v) for v in args.size.lower().split("x"))

    rank, local_rank, world = parallel.init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.conv_pipeline is not None:
        from tdnet_amd import _capi
        _capi.lib().tdnet_set_conv_pipeline(args.conv_pipeline)
    if args.precision == "fp16":
        from tdnet_amd import _capi
        _capi.lib().tdnet_set_conv_precision(1)
    if args.winograd is not None:
        from tdnet_amd import _capi
        _capi.lib().tdnet_set_conv_winograd(args.winograd)
    if args.backbone is None:
        args.backbone = "resnet101" if args.model == "psp" else "resnet18"
    spec = arch.model_spec(args.model, 19, args.backbone)
    P = spec.path_num
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = weights.synth_state_dict(spec, h, w, 0) if rank == 0 else None
    sd = parallel.broadcast_state_dict(spec, h, w, sd, dev)                       # the one RCCL collective on the data path
    if args.model == "psp":                                                       # the reference's comparison model (test.py:34-38)
        model = pspnet.pspnet(nclass=19, model_path=None, backbone=args.backbone).eval().to(dev)
    else:
        cls = td4_psp18.td4_psp18 if args.model == "td4" else td2_psp50.td2_psp50
        model = cls(nclass=19, path_num=P, model_path=None, backbone=args.backbone).eval().to(dev)
    model.load_state_dict(sd)
    C = max(1, args.clips_per_gpu)
    models, streams = [model], [torch.cuda.current_stream(dev)]
    for _ in range(C - 1):                                                        # extra clips: own handle (weights + FIFO), own stream
        m2 = type(model)(nclass=19, model_path=None, backbone=args.backbone).eval().to(dev) if args.model == "psp" else \
            type(model)(nclass=19, path_num=P, model_path=None, backbone=args.backbone).eval().to(dev)
        m2.load_state_dict(sd)
        models.append(m2); streams.append(torch.cuda.Stream(dev))

    # one clip per handle (different seeds), pre-staged on the device; frames cycle, pos_id keeps counting
    NF = 8
    clips = [[torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100 + rank + 1000 * c)] for c in range(C)]
    clip = clips[0]
    t_frame = 0

    pp = parallel.PathParallelStream(model, P, rank=rank, world=world, device=dev) if args.mode == "path-parallel" else None
    if pp is not None:
        clip = clips[0] = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100)]   # the SAME stream on every rank
        C = 1

    def step():
        nonlocal t_frame
        if pp is not None:                                            # one round: frames t_frame .. t_frame + world - 1
            # PathParallelStream.process keeps pos_id = t mod P only if rounds start at multiples of lcm(world, P): NF = 8 does
            out = pp.process([clip[(t_frame + j) % NF] for j in range(world)], first_frame=t_frame)
            t_frame += world
            return out
        out = None
        for c in range(C):
            with torch.cuda.stream(streams[c]):
                o = models[c](clips[c][t_frame % NF], pos_id=t_frame % P)
            out = o if out is None else out
        t_frame += 1
        return out

    with torch.no_grad():
        for _ in range(max(args.warmup, P + 2)):
            step()
        torch.cuda.synchronize(dev)
        parallel.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        parallel.barrier()
        dt = time.perf_counter() - t0
    tmax = parallel.allreduce_max(torch.tensor([dt], dtype=torch.float64, device=dev)).item()
    fps = world * C * args.steps / tmax

    mname = ("psp%s" if args.model == "psp" else args.model + "-psp%s") % args.backbone[6:]
    res = {"metric": "frames/sec (%s, %dx%d, full-resolution logits)" % (mname, H, W),
           "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, P + 2),
           "ms_per_step": round(1e3 * tmax / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if args.precision == "fp32" else "f16 conv operands (fp32 accumulate, fp32 storage; attention/LN/PPM fp32)",
           "data": "synthetic",
           "config": {"workload": "%s, %dx%d Cityscapes-shaped synthetic stream, %d-frame feature cache, %d clip%s per GPU"
                                  % (mname, H, W, spec.fifo, C, "" if C == 1 else "s (concurrent HIP streams)"),
                      "parallelism": ("clip-parallel x%d, RCCL weight broadcast only" % world) if pp is None else
                                     ("path-parallel x%d: one stream, one all-gather of %d cache entries per round" % (world, world)),
                      "target_fps_per_gpu": 30}}
    if pp is not None:
        res["scaling"] = "strong"

    if rank == 0:
        eng = model.engine
        gflop = eng.flops_per_frame() / 1e9
        res["config"]["algorithmic_gflop_per_frame"] = round(gflop, 1)
        res["config"]["frame_tflops"] = round(gflop * (C * args.steps / tmax) / 1e3, 2)
        # ---- roofline of the dominant kernel: profiled replay (HIP events around every launch, same stream) ----------
        eng.set_profiling(True)
        acc = {k: [0.0, 0.0, 0.0] for k in (0, 1, 2, 3)}
        nprof = 2 * P
        C_timed, C = C, 1                                             # the replay runs clip 0 alone: per-launch durations, no co-running streams
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for _ in range(nprof):
                step()
                torch.cuda.synchronize(dev)
                for k in acc:
                    ms, fl, n = eng.last(k)
                    acc[k][0] += ms; acc[k][1] += fl; acc[k][2] += n
        eng.set_profiling(False)
        dom_ms, dom_fl, dom_n = acc[3]
        if dom_n > 0 and dom_ms > 0:
            achieved = dom_fl / (dom_ms * 1e-3) / 1e12
            from tdnet_amd import _capi
            cfgbits = _capi.lib().tdnet_get_conv_config()
            if cfgbits & 2:
                kname, peak = "k_conv_igemm_h<128,128,2,2,3> (3x3 dilated conv, fp16-input MFMA, fp32 accumulate)", 2500.0
            elif (cfgbits >> 2) & 7:
                gk = "k_gemm_persistent" if (cfgbits >> 5) & 1 else "k_conv_igemm<.,.,.,.,1>"
                f4 = ((cfgbits >> 2) & 7) >= 3
                kname, peak = (gk + " x%d (batched GEMM of the Winograd F(%s,3x3) convs of layers %s + head, "
                               "fp32 MFMA; FLOP = executed GEMM FLOP, %sx fewer than the direct conv's)"
                               % ((36, "4x4", "2-4", "4") if f4 else (16, "2x2", "3-4", "2.25"))), PEAK_FP32_MFMA_TFLOPS
            else:
                kname, peak = "k_conv_igemm<128,128,2,2,3> (3x3 dilated conv, fp32 MFMA)", PEAK_FP32_MFMA_TFLOPS
            res["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(achieved / peak, 4), "traffic": traffic_from_profiles(),
                               "avg_launch_ms": round(dom_ms / dom_n, 4), "launches_per_frame": dom_n / nprof,
                               "gflop_per_launch": round(dom_fl / dom_n / 1e9, 2)}
        res["breakdown_ms_per_frame"] = {"conv_gemm": round(acc[0][0] / nprof, 3), "attention": round(acc[1][0] / nprof, 3),
                                         "hbm_bound_tail": round(acc[2][0] / nprof, 3)}
        res["breakdown_tflops"] = {"conv_gemm": round(acc[0][1] / max(acc[0][0], 1e-9) / 1e9, 2),
                                   "attention": round(acc[1][1] / max(acc[1][0], 1e-9) / 1e9, 2)}

        # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only) ------------------------------------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import tdnet_ref                                          # checker / baseline only
            cores = tdnet_ref.tune_threads()          # threads actually used (fastest of 8..128 on a probe conv)
            ref = (tdnet_ref.PSPNetRef if args.model == "psp" else tdnet_ref.TDNetRef)(spec, sd)
            model.reset()
            nwarm, nsteady = P, max(1, args.cpu_frames)
            cpu_t, worst, flips, npx = 0.0, 0.0, 0, 0
            hist = np.zeros((19, 19), np.int64)
            with torch.no_grad():
                for t in range(nwarm + nsteady):
                    x = clip[t % NF]
                    out = model(x, pos_id=t % P).cpu()
                    xc = x.cpu()
                    c0 = time.perf_counter()
                    exp = ref.forward(xc, t % P)
                    c1 = time.perf_counter()
                    if t >= nwarm:
                        cpu_t += c1 - c0
                    worst = max(worst, (out - exp).abs().max().item())
                    lo, lr = out[0].argmax(0).numpy(), exp[0].argmax(0).numpy()
                    flips += int((lo != lr).sum()); npx += lo.size
                    hist += tdnet_ref.confusion_miou(lo, lr, 19)[1]
            iu = np.diag(hist) / np.maximum(1, hist.sum(1) + hist.sum(0) - np.diag(hist))
            res["cpu_baseline"] = {"value": round(nsteady / cpu_t, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": "%d steady-state frames of the same clip (after %d warm-up frames), oracle/tdnet_ref.py "
                                             "= the reference's op graph on torch-CPU %s with %d threads (host has %d)"
                                             % (nsteady, nwarm, torch.__version__, cores, os.cpu_count() or 1)}
            res["parity"] = {"frames": nwarm + nsteady, "max_abs_dlogit": float("%.3e" % worst), "label_mismatches": flips,
                             "pixels": npx, "miou_vs_cpu": round(float(iu[hist.sum(1) > 0].mean()), 6)}
        print(json.dumps(res), flush=True)
    parallel.barrier()


if __name__ == "__main__":
    main()
