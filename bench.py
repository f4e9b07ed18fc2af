#!/usr/bin/env python3
"""bench.py -- frames/s of the TDNet per-frame hot path on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model td4|td2|psp] [--size HxW]

A "step" is one frame of one synthetic video clip through model(image, pos_id) (Testing/test.py:53): sub-network
forward + attention propagation from the cached frames + head + x8 upsample to full-resolution logits.  Frames are
pre-staged in HBM (the timed region holds no H2D copy); every rank serves its own clip (weak scaling, no per-frame
collective); weights are generated on rank 0 and broadcast once over RCCL.  Timing: barrier + synchronize, EXACTLY K
steps, synchronize + barrier, max over ranks.

N ranks: under `python -m torch.distributed.run ...` (WORLD_SIZE set) this process is one rank.  Called plainly with
--gpus N > 1 it LAUNCHES the N ranks itself (re-executes under torch.distributed.run on 127.0.0.1) and fails if the
node has fewer than N GPUs: `python bench.py --gpus 8` never quietly measures one GPU.  Every line carries
`world_size_seen` (an all-reduce of ones), `rccl_bcast_ms` and the per-rank frames/s.

Extra objects on the JSON line:
  roofline      the dominant kernel: achieved = executed FLOP per launch / average launch duration (HIP events on the forward's
                stream during a profiled replay right after the timed region); peak = 157.3 TFLOP/s fp32 MFMA
                (MI355X_MICROARCH.md); traffic = HBM-side bytes per launch of that kernel SYMBOL, measured in this run by two
                child passes of this script under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; one counter per pass, FETCH x2 per the
                gfx950 calibration), never read from a file.
  frame         frame-level accounting: algorithmic FLOP (the reference's op list) and EXECUTED FLOP (Winograd GEMMs do 1/4 of the
                direct convs' MACs) per frame, both as TFLOP/s and as fractions of the fp32 roof, total HBM bytes per frame over all
                kernels from the same PMC passes, and the all-direct-convolution configuration timed beside the headline.
  cpu_baseline  the CPU oracle (oracle/tdnet_ref.py, the reference's op graph on torch-CPU/oneDNN) on a bounded sample.
  parity        GPU logits vs that oracle on the sampled frames, with the tie-band rule of the GPU tests: a label may differ only
                where the reference's top-2 gap is <= 2 max|dlogit|; `flips_outside_tie_band` > 0 or max|dlogit| > 1e-3 makes the
                fp32 run exit non-zero (fp16 mode: max|dlogit| > 3e-2, < 99.5 % of the labels equal or mIoU < 0.99 does).
  sustained     `value` is EXACTLY --steps frames between two barriers (the driver's contract); when that window is shorter than
                0.5 s the same loop is timed again for as many frames as fill >= 0.5 s and reported beside it (steps_timed, value).
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # before the HIP runtime starts (tdnet_amd/__init__.py says why: 275 -> 185 frames/s behind RCCL otherwise)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_FP16_MFMA_TFLOPS = 2500.0
PEAK_BF16X3_TFLOPS = 2500.0 / 6.0      # precision 2: six bf16-MFMA products per fp32 product (td_gemm_b3.h): the dense bf16 MFMA peak / 6 = 416.7


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--model", default="td4", choices=["td4", "td2", "psp"])
    ap.add_argument("--backbone", default=None, help="resnet18 (default) | resnet34 | resnet50 (td2) | resnet101 (psp)")
    ap.add_argument("--size", default="1024x2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-direct-line", action="store_true", help="skip timing the all-direct-conv configuration beside the headline")
    ap.add_argument("--cpu-frames", type=int, default=5, help="steady-state frames timed on the CPU oracle (SURVEY 8d: >= 5)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs for BASELINE configs[1] (td2-psp18 1024x2048 fp32) and configs[4] (td2-psp34 720x960 fp16, the "
                         "stand-in for the BiSeNet-34 the reference does not contain) that the default N = 1 run appends as `other_configs`")
    ap.add_argument("--quick", action="store_true", help="= --no-cpu-baseline --no-pmc --no-direct-line --no-other-configs, no `sustained` window (A/B and profiler runs)")
    ap.add_argument("--perturb-rank", type=int, default=-1,
                    help="TEST ONLY: this rank scales one weight tensor by 1.001 after the broadcast; the N-rank line must then say "
                         "ranks_agree false and the run must exit non-zero (tests/test_bench_launch.py)")
    ap.add_argument("--conv-pipeline", type=int, default=None, help="tuning: 0 one-stage / 1 two-stage conv prefetch")
    ap.add_argument("--winograd", type=int, default=None, help="conv algorithm: 0 direct, 3 Winograd F(4x4,3x3) for layers 2-4 + head (default: library default), 4 F(4x4,3x3) for every stride-1 3x3")
    ap.add_argument("--attention", type=int, default=None, help="0 exact two-pass softmax, 1 single pass (lazily moved reference), 2 the same with one barrier per key tile (library default)")
    ap.add_argument("--fusion", type=int, default=None, help="bit mask of launch-level fusions (include/tdnet.h tdnet_opts.fusion)")
    ap.add_argument("--gemm-persistent", type=int, default=None, help="tuning: 1 persistent GEMM on a full wave of workgroups (default), 0 one tile per workgroup, n > 1 grid forced to n")
    ap.add_argument("--overlap", type=int, default=None, help="bit mask (include/tdnet.h tdnet_opts.overlap): 1 = layers 3-4 as two row-parity chains on two streams, 2 = low-register Winograd transforms everywhere, bits 4-5 = channels per lane")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16x3"],
                    help="fp32 (default; the mode the parity gate is defined for) | fp16 = fp16 MFMA, fp32 accumulate (BASELINE "
                         "config 5); parity vs the fp32 CPU path is gated at 3e-2 / 99.5 % of the labels / mIoU 0.99 (tests/test_gpu_fp16.py) | "
                         "bf16x3 = tdnet_opts.precision 2: fp32 operands split three ways into bf16, six products on the bf16 MFMA, fp32 "
                         "accumulate (fp32-accurate, opt-in; held to the fp32 gate)")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="independent clips served concurrently by one GPU, each with its own handle/FIFO on its own HIP stream "
                         "(throughput mode; a step is then one frame of EVERY clip).  Default 1 = BASELINE's one clip per GPU")
    ap.add_argument("--mode", default="clips", choices=["clips", "path-parallel", "frame-pipelined"],
                    help="clips (default, BASELINE): independent clips, one per GPU, no per-frame communication | path-parallel: ONE "
                         "stream served by all N ranks, one all-gather of cache entries per round of N frames (a step = one round) | "
                         "frame-pipelined: each rank's ONE clip with two frames in flight (two handles on two HIP streams of the rank, "
                         "parallel.FramePipelinedStream; a step = one round of two frames; throughput mode for maps that leave CUs idle)")
    ap.add_argument("--graph", action="store_true",
                    help="the timed loop replays ONE hipGraph per pos_id cycle (P frame calls captured on a stream in steady state: tdnet_amd/graph.py) "
                         "instead of enqueueing every frame's launches; a step is then one cycle of P frames.  The default run reports the same comparison "
                         "as the `hipgraph` object of the line without changing `value`")
    ap.add_argument("--no-numerics-stress", action="store_true", help="skip the un-calibrated-init numerics leg (`numerics_stress`: fp64 oracle on 4 frames at 769x1537)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check WITHOUT the model (CPU, gloo): rank launch, rendezvous, weight broadcast, barriers, timing "
                         "reduction and the JSON line; `value` is null.  Used by tests/test_bench_launch.py; never a measurement")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 and the collectives run over gloo, so the N-rank code path (launch, weight "
                         "broadcast into N model instances, barriers, reductions) can be exercised with the real kernels on a box with "
                         "ONE GPU; the line says shared_gpu and its value is null")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # this process runs under rocprofv3 for a counter pass
    a = ap.parse_args(argv)
    if a.quick:
        a.no_cpu_baseline = a.no_pmc = a.no_direct_line = a.no_other_configs = a.no_numerics_stress = True
    return a


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(args, argv):
    """`python bench.py --gpus N` without a launcher: become the launcher.  One rank per GPU, rendezvous on 127.0.0.1."""
    if not args.dry_run:
        import torch
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if args.share_gpu and n >= 1:
            n = args.gpus
        if n < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s); refusing to report a smaller job under that label\n" % (args.gpus, n))
            sys.exit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.call(cmd, env=env))


# ---- live HBM-traffic measurement: child passes of this script under rocprofv3 --pmc ---------------------------------------
def find_rocprof():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def pmc_pass(counter, child_args, timeout=420):
    """One counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Returns
    ({kernel name: [sum of counter values, dispatches]}, error string or None)."""
    prof = find_rocprof()
    if prof is None:
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="tdnet_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.abspath(__file__)] + child_args
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stdout.decode(errors="replace")[-300:])
        agg = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                a = agg.setdefault(row["Kernel_Name"], [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row.get("Dispatch_Id", len(a[1])))
        return {k: [v[0], len(v[1])] for k, v in agg.items()}, None
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 --pmc %s timed out" % counter
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_traffic(child_args, dom_regex, frames):
    """(bytes per launch of the dominant kernel symbol, bytes per frame over all kernels, detail dict) from separate FETCH_SIZE /
    WRITE_SIZE passes.  Counter values are KB; FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 (guide, HBM
    section), WRITE_SIZE is used as reported."""
    fetch, e1 = pmc_pass("FETCH_SIZE", child_args)
    if fetch is None:
        return None, None, {"traffic_source": "unavailable: " + e1}
    write, e2 = pmc_pass("WRITE_SIZE", child_args)
    if write is None:
        return None, None, {"traffic_source": "unavailable: " + e2}
    rx = re.compile(dom_regex)
    f_sum = sum(v[0] for k, v in fetch.items() if rx.search(k))
    f_n = sum(v[1] for k, v in fetch.items() if rx.search(k))
    w_sum = sum(v[0] for k, v in write.items() if rx.search(k))
    w_n = sum(v[1] for k, v in write.items() if rx.search(k))
    if not f_n or not w_n:
        return None, None, {"traffic_source": "unavailable: no dispatch of /%s/ in the counter passes" % dom_regex}
    per_launch = (2.0 * f_sum / f_n + w_sum / w_n) * 1024.0
    total = (2.0 * sum(v[0] for v in fetch.values()) + sum(v[0] for v in write.values())) * 1024.0
    detail = {"traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run (2 x FETCH_SIZE + WRITE_SIZE)",
              "traffic_kernel_symbols": sorted(k.split("(")[0].replace("void ", "") for k in fetch if rx.search(k)),
              "traffic_dispatches": f_n,
              "fetch_bytes_per_launch_x2": round(2.0 * f_sum / f_n * 1024.0), "write_bytes_per_launch": round(w_sum / w_n * 1024.0)}
    return per_launch, total / frames, detail


# ---- parity of a model against the CPU oracle on a bounded sample (the rule of tests/test_gpu_model.py::check_frame) ---------------
def parity_sample(model, ref, clip, P, nw, nsteady, fp32, torch, np, tdnet_ref, keep_labels=None):
    """Runs frames 0 .. nw + nsteady - 1 of `clip` through the GPU model (after reset) and the oracle.  Returns (parity dict, CPU
    seconds spent on the nsteady steady-state frames, 19x19 confusion matrix GPU labels vs oracle labels)."""
    import time as _t
    model.reset()
    cpu_t, worst, flips, outside, npx = 0.0, 0.0, 0, 0, 0
    hist = np.zeros((19, 19), np.int64)
    with torch.no_grad():
        for t in range(nw + nsteady):
            x = clip[t % len(clip)]
            out = model(x, pos_id=t % P).cpu()
            xc = x.cpu()
            c0 = _t.perf_counter()
            exp = ref.forward(xc, t % P)
            c1 = _t.perf_counter()
            if t >= nw:
                cpu_t += c1 - c0
            err = (out - exp).abs().max().item()
            worst = max(worst, err)
            lo, lr = out[0].argmax(0).numpy(), exp[0].argmax(0).numpy()
            if keep_labels is not None:
                keep_labels.append(lr.astype(np.uint8))
            bad = lo != lr
            flips += int(bad.sum()); npx += lo.size
            if bad.any():
                top2 = np.sort(exp[0].numpy(), axis=0)[-2:]
                # fp32: the band of the clip's worst error (the gate of the GPU tests); fp16 mode: twice THIS pixel's own logit error
                band = 2 * err if fp32 else 2.0 * (out[0] - exp[0]).abs().max(0)[0].numpy()[bad]
                outside += int(((top2[1] - top2[0])[bad] > band).sum())
            hist += tdnet_ref.confusion_miou(lo, lr, 19)[1]
    iu = np.diag(hist) / np.maximum(1, hist.sum(1) + hist.sum(0) - np.diag(hist))
    par = {"frames": nw + nsteady, "max_abs_dlogit": float("%.3e" % worst), "label_mismatches": flips,
           "flips_outside_tie_band": outside, "pixels": npx, "miou_vs_cpu": round(float(iu[hist.sum(1) > 0].mean()), 6),
           "min_class_iou_vs_cpu": round(float(iu[hist.sum(1) > 0].min()), 6),
           "labels_equal_frac": round(1.0 - flips / max(1, npx), 6),
           "gate": "max|dlogit| <= 1e-3 and every label flip inside the reference's top-2 tie band (gap <= 2 max|dlogit|)"
                   if fp32 else "fp16 mode, the gate of tests/test_gpu_fp16.py: max|dlogit| <= 3e-2, >= 99.5 % of the labels equal, mIoU >= 0.99, every class's IoU >= 0.97, "
                                "label flips only where the reference's top-2 gap <= 2 x that pixel's own |dlogit| (tie band per pixel)"}
    if fp32 and (outside > 0 or worst > 1e-3):
        par["FAILED"] = True
    if not fp32 and (worst > 3e-2 or par["labels_equal_frac"] < 0.995 or par["miou_vs_cpu"] < 0.99 or par["min_class_iou_vs_cpu"] < 0.97 or outside > 0):
        par["FAILED"] = True
    return par, cpu_t, hist


def logits_digest(out, torch):
    """Two 64-bit integers of a logits tensor's BIT pattern (plain sum and a position-weighted sum, int64 wrap-around): equal on two
    ranks iff -- up to a 2^-64-ish accident -- the tensors are bit-identical."""
    v = out.contiguous().view(torch.int32).flatten().to(torch.int64)
    idx = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
    return [int(v.sum().item()), int((v * idx).sum().item())]


def profile_dominant(eng, step_one, nprof, sync, torch):
    """Profiled replay (HIP events around every launch on the forward's own streams): per-family [ms, FLOP, launches] sums."""
    eng.set_profiling(True)
    acc = {k: [0.0, 0.0, 0.0] for k in (0, 1, 2, 3, 4)}
    sync()
    with torch.no_grad():
        for _ in range(nprof):
            step_one()
            sync()
            for k in acc:
                ms_, fl, n = eng.last(k)
                acc[k][0] += ms_; acc[k][1] += fl; acc[k][2] += n
    eng.set_profiling(False)
    return acc


def latency_synced(step_one, sync, nframes=20, skip=6):
    """The reference's OWN measurement (Testing/test.py:46-59): synchronise, start the clock, one forward, synchronise, stop; the frames
    i > 5 are averaged.  A per-frame LATENCY with the host in the loop, not a throughput: the partner of BASELINE.md's published
    ms/frame figures.  Returns milliseconds per frame."""
    tot, cnt = 0.0, 0
    for i in range(nframes + skip):
        sync()
        t0 = time.perf_counter()
        step_one()
        sync()
        dt = time.perf_counter() - t0
        if i >= skip:
            tot += dt
            cnt += 1
    return 1e3 * tot / max(cnt, 1)


DTYPE_BF16X3 = "f32 via bf16x3 split, fp32 accumulate (GEMM operands as three bf16 parts, six products on the bf16 MFMA; tdnet_opts.precision = 2, opt-in)"


def graph_compare(m, clip, P, NF, H, W, dev, sync, steps, torch):
    """The clip's steady state as ONE hipGraph per pos_id cycle (tdnet_amd/graph.py GraphedClip: P frame calls captured on a stream, replayed)
    against the eager loop on the same model: frames/s, host time per frame to enqueue (measured on an empty queue) and bit identity of the
    replayed frames.  The model is left in step with `clip` at a multiple of P frames."""
    from tdnet_amd.graph import GraphedClip
    cycles = max(2, steps // P)
    with torch.no_grad():
        m.reset()
        t = 0
        for _ in range(2 * P):                                             # steady state, aligned to the cycle
            m(clip[t % NF], pos_id=t % P); t += 1
        eager = [m(clip[(t + j) % NF], pos_id=(t + j) % P).clone() for j in range(P)]
        sync()
        th = time.perf_counter()
        for j in range(P):                                                 # host cost of one eager cycle on an empty queue
            m(clip[(t + P + j) % NF], pos_id=(t + P + j) % P)
        host_eager = (time.perf_counter() - th) / P * 1e6
        sync()
        t0 = time.perf_counter()
        tt = t + 2 * P
        for _ in range(cycles * P):
            m(clip[tt % NF], pos_id=tt % P); tt += 1
        sync()
        eager_fps = cycles * P / (time.perf_counter() - t0)
        # the same frames through the graph: rewind to the state before `eager` was computed
        m.reset()
        for u in range(t):
            m(clip[u % NF], pos_id=u % P)
        g = GraphedClip(m, H, W, dev)
        outs = g.replay([clip[(t + j) % NF] for j in range(P)])
        sync()
        same = all(torch.equal(a, b) for a, b in zip(outs, eager))
        sync()
        th = time.perf_counter()
        g.replay([clip[(t + P + j) % NF] for j in range(P)])
        host_graph = (time.perf_counter() - th) / P * 1e6
        sync()
        staged = [[clip[(t + c * P + j) % NF] for j in range(P)] for c in range(2)]
        t0 = time.perf_counter()
        for c in range(cycles):
            g.replay(staged[c & 1])
        sync()
        graph_fps = cycles * P / (time.perf_counter() - t0)
        m.reset()
    return {"what": "P = %d consecutive frame calls captured into one hipGraph on a caller stream (torch.cuda.CUDAGraph) in steady state and replayed per cycle; "
                    "inputs copied into the graph's static frame buffers (included in the graph figures)" % P,
            "eager_fps": round(eager_fps, 2), "graph_fps": round(graph_fps, 2), "graph_vs_eager": round(graph_fps / eager_fps, 4),
            "host_launch_us_per_frame_eager": round(host_eager, 1), "host_launch_us_per_frame_graph": round(host_graph, 1),
            "frames_timed": cycles * P, "bit_identical_to_eager": bool(same)}


def other_config_leg(tag, model_name, backbone, size, precision, steps, cpu_frames, dev, sync, cite, pipelined=False, published_ms=None, graph=False):
    """One short leg for another BASELINE.json config on this GPU: frames/s over `steps` steady frames, the dominant kernel's roofline
    fraction (profiled replay) and parity against the CPU oracle on `cpu_frames` steady frames.  Own model, own clip, own oracle."""
    import numpy as np
    import torch
    from oracle import tdnet_ref                                          # checker only
    from tdnet_amd import arch, weights
    from tdnet_amd.model import pspnet, td2_psp50, td4_psp18
    H, W = size
    spec = arch.model_spec(model_name, 19, backbone)
    P = spec.path_num
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    cls = td4_psp18.td4_psp18 if model_name == "td4" else td2_psp50.td2_psp50
    opts = {"precision": 1} if precision == "fp16" else {"precision": 2} if precision == "bf16x3" else {}
    if model_name == "psp":
        m = pspnet.pspnet(nclass=19, model_path=None, backbone=backbone, kernel_opts=opts).eval().to(dev)
    else:
        m = cls(nclass=19, path_num=P, model_path=None, backbone=backbone, kernel_opts=opts).eval().to(dev)
    m.load_state_dict(sd)
    NF = 6
    clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=200)]
    st = {"t": 0}

    def step():
        t = st["t"]
        m(clip[t % NF], pos_id=t % P)
        st["t"] = t + 1
    with torch.no_grad():
        for _ in range(max(12, P + 2)):                                # the first frames of a new model also load its kernel variants
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0
    eng = m.engine
    peak = PEAK_FP16_MFMA_TFLOPS if precision == "fp16" else PEAK_BF16X3_TFLOPS if precision == "bf16x3" else PEAK_FP32_MFMA_TFLOPS
    with torch.no_grad():
        lat_ms = latency_synced(step, sync)
    nlaunch = eng.last_launch_count()
    wbytes, hbytes, _ = eng.memory_bytes()
    acc = profile_dominant(eng, step, 2 * P, sync, torch)
    dom_ms, dom_fl, dom_n = acc[3]
    mname = ("psp%s" if model_name == "psp" else model_name + "-psp%s") % backbone[6:]
    leg = {"config": tag, "workload": "%s, %dx%d, %s" % (mname, H, W, precision), "reference": cite,
           "value": round(steps / dt, 3), "unit": "frames/s", "steps": steps, "ms_per_step": round(1e3 * dt / steps, 4),
           "latency_ms_synced": round(lat_ms, 4),
           "dtype": "f32" if precision == "fp32" else DTYPE_BF16X3 if precision == "bf16x3" else "f16 (fp16 MFMA, fp32 accumulate)", "kernel_opts": eng.opts(),
           "algorithmic_gflop": round(eng.flops_per_frame() / 1e9, 1), "launches_per_frame": nlaunch,
           "memory": {"weights_bytes": wbytes, "handle_bytes": hbytes}}
    if published_ms:
        leg["published"] = {"ms_per_frame": published_ms, "hardware": "Titan Xp, PyTorch 1.1.0 + CUDA 10.0 (Testing/TEST_README.md:31-33)", "protocol": "test.py:46-59",
                            "speedup_of_latency_ms_synced": round(published_ms / lat_ms, 2),
                            "note": "other hardware, trained checkpoint vs synthetic weights of the same architecture: orientation, not vs_baseline"}
    if dom_n > 0 and dom_ms > 0:
        leg["roofline"] = roofline_block(acc, precision == "fp16", peak, 2 * P)
        if precision == "bf16x3":
            leg["roofline"]["kernel"] = ("k_gemm_b3<1> (the Winograd F(4x4) GEMMs with fp32 operands as three bf16 parts: six bf16-MFMA products per fp32 product, fp32 "
                                         "accumulate; td_gemm_b3.h), executed GEMM FLOP")
            leg["roofline"]["peak_note"] = "2500 TFLOP/s dense bf16 MFMA / 6 products"
    if graph:
        leg["hipgraph"] = graph_compare(m, clip, P, NF, H, W, dev, sync, steps, torch)
        if not leg["hipgraph"]["bit_identical_to_eager"]:
            leg.setdefault("parity", {})["FAILED"] = True
    if cpu_frames > 0:
        ref = (tdnet_ref.PSPNetRef if model_name == "psp" else tdnet_ref.TDNetRef)(spec, sd)
        par, cpu_t, _ = parity_sample(m, ref, clip, P, P, cpu_frames, precision != "fp16", torch, np, tdnet_ref)
        leg["parity"] = par
        leg["cpu_baseline"] = {"value": round(cpu_frames / cpu_t, 4), "unit": "frames/s", "kind": "port", "sample": "%d steady-state frames" % cpu_frames}
    if pipelined:
        # the same clip with TWO FRAMES IN FLIGHT (parallel.FramePipelinedStream: two handles on two HIP streams, the cache entry handed
        # over between encode and propagate): first held bit for bit to this leg's single handle on a replay from an empty FIFO, then timed
        from tdnet_amd import parallel
        nchk = P + 4
        with torch.no_grad():
            m.reset()
            refs = [m(clip[t % NF], pos_id=t % P).clone() for t in range(nchk)]
            sync()
            m.reset()
            t_l0 = time.perf_counter()
            fp = parallel.FramePipelinedStream.from_model(m, P, dev, (H, W))      # lane 1 shares lane 0's weight block (tdnet_create_shared)
            sync()
            lane_s = time.perf_counter() - t_l0
            stages = fp.stages
            lane_w, lane_h, lane_refs = stages[1].engine.memory_bytes()
            outs = fp.process([clip[t % NF] for t in range(nchk)])
            same = all(torch.equal(x, y) for x, y in zip(outs, refs))
            del outs, refs
            t = nchk                                                    # even: the rounds stay aligned with pos_id = t mod P
            for _ in range(6):
                fp.process([clip[t % NF], clip[(t + 1) % NF]], first_frame=t, join=False); t += 2
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):                                      # `steps` rounds = 2 x steps frames
                fp.process([clip[t % NF], clip[(t + 1) % NF]], first_frame=t, join=False); t += 2
            sync()
            dt2 = time.perf_counter() - t0
        leg["two_frames_in_flight"] = {"value": round(2 * steps / dt2, 3), "unit": "frames/s", "frames": 2 * steps, "vs_one_handle": round(2 * steps / dt2 / leg["value"], 3),
                                       "bit_identical_to_one_handle": bool(same), "frames_compared": nchk,
                                       "second_lane": {"shares_weight_block": lane_refs == 2, "extra_hbm_bytes": lane_h, "weight_block_bytes_shared": lane_w,
                                                       "create_s": round(lane_s, 3)},
                                       "what": "ONE clip, two handles on two HIP streams of this process, frame t + 1 encoded beside frame t (bench.py --mode frame-pipelined); "
                                               "throughput of small maps that leave CUs idle, a frame's latency grows"}
        if not same:
            leg["two_frames_in_flight"]["FAILED"] = True
            leg.setdefault("parity", {})["FAILED"] = True
        for st_ in stages:
            st_._close_engines()
        del stages, fp
    else:
        eng.close()
    del m
    return leg


def numerics_stress(dev, torch, np, frames=4, H=769, W=1537):
    """Numerics at REAL logit magnitudes on the line: td4-psp18 at the checkpoint's geometry (769x1537, [97,193] LayerNorm affine, td4_psp18.py:107-110)
    with SURVEY 8d's UN-CALIBRATED init (every conv ~ N(0, 2 / (k k C_out)), resnet.py:162-165: logits in the tens), `frames` frames (the last one in
    steady state), against an fp64 evaluation of the oracle graph ("truth") and the fp32 CPU oracle.  The gate of
    tests/test_gpu_model.py::test_reference_init_at_the_checkpoints_geometry_769x1537: max|gpu - truth| <= 4x and rms <= 3x the fp32 CPU path's own,
    max|gpu - cpu| <= 3x the CPU's own error, a label differs from the truth's only inside the truth's top-2 tie band.  One CPU evaluation, three GPU
    configurations: the default kernels, all-direct convolutions (winograd 0) and precision 2 (bf16x3 split GEMMs)."""
    from oracle import tdnet_ref                                          # checker only
    from tdnet_amd import arch, weights
    from tdnet_amd.model import td4_psp18
    spec = arch.model_spec("td4", 19, "resnet18")
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0, init="reference")
    ref32 = tdnet_ref.TDNetRef(spec, sd)
    ref64 = tdnet_ref.TDNetRef(spec, {k: torch.from_numpy(np.asarray(v)).double() for k, v in sd.items()})
    clip = weights.synth_video(H, W, frames, seed=1)
    t0 = time.perf_counter()
    truth, cpu = [], []
    for t, x in enumerate(clip):
        xt = torch.from_numpy(x)
        truth.append(ref64.forward(xt.double(), t % 4).numpy())
        cpu.append(ref32.forward(xt, t % 4).double().numpy())
    cpu_s = time.perf_counter() - t0
    edges = np.array([0.0, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, np.inf])
    e_cpu = max(float(np.abs(c - tr).max()) for c, tr in zip(cpu, truth))
    n = sum(tr.size for tr in truth)
    r_cpu = (sum(float(((c - tr) ** 2).sum()) for c, tr in zip(cpu, truth)) / n) ** 0.5
    out = {"workload": "td4-psp18 %dx%d, un-calibrated (reference) init, %d frames; truth = the oracle graph in fp64" % (H, W, frames),
           "max_abs_truth": round(max(float(np.abs(tr).max()) for tr in truth), 2), "cpu_fp32_max_err": float("%.3e" % e_cpu), "cpu_fp32_rms_err": float("%.3e" % r_cpu),
           "oracle_seconds": round(cpu_s, 1),
           "gate": "max err <= 4x and rms <= 3x the fp32 CPU path's own distance to the fp64 truth; max|gpu - cpu| <= 3x that distance; label flips only inside the truth's top-2 tie band",
           "top2_gap_decades": "[0,1e-4) [1e-4,1e-3) [1e-3,1e-2) [1e-2,1e-1) [1e-1,1) [1,inf)", "configs": {}}
    for tag, opts in (("default", {}), ("winograd 0 (all-direct convs)", {"winograd": 0}), ("precision 2 (bf16x3 split GEMMs)", {"precision": 2})):
        m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, kernel_opts=opts).eval().to(dev)
        m.load_state_dict(sd)
        e_gpu, s_gpu, e_gc, outside = 0.0, 0.0, 0.0, 0
        gap_hist, flip_hist = np.zeros(6, np.int64), np.zeros(6, np.int64)
        with torch.no_grad():
            for t, x in enumerate(clip):
                o = m(torch.from_numpy(x).to(dev), pos_id=t % 4).cpu().double().numpy()
                dg = o - truth[t]
                e_gpu = max(e_gpu, float(np.abs(dg).max())); s_gpu += float((dg ** 2).sum())
                e_gc = max(e_gc, float(np.abs(o - cpu[t]).max()))
                top2 = np.sort(truth[t][0], axis=0)[-2:]
                gap = top2[1] - top2[0]
                bad = o[0].argmax(0) != truth[t][0].argmax(0)
                gap_hist += np.histogram(gap, edges)[0]
                flip_hist += np.histogram(gap[bad], edges)[0]
                outside += int((gap[bad] > 2 * float(np.abs(dg).max())).sum())
        m.engine.close()
        del m
        r_gpu = (s_gpu / n) ** 0.5
        c = {"max_err": float("%.3e" % e_gpu), "rms_err": float("%.3e" % r_gpu), "max_err_vs_cpu_own": round(e_gpu / e_cpu, 3), "rms_err_vs_cpu_own": round(r_gpu / r_cpu, 3),
             "max_abs_gpu_minus_cpu": float("%.3e" % e_gc), "gpu_minus_cpu_vs_cpu_own": round(e_gc / e_cpu, 3),
             "pixels_per_gap_decade": gap_hist.tolist(), "labels_differing_from_truth_per_gap_decade": flip_hist.tolist(), "flips_outside_tie_band": outside}
        if e_gpu > 4.0 * e_cpu or r_gpu > 3.0 * r_cpu or e_gc > 3.0 * e_cpu or outside > 0:
            c["FAILED"] = True
            out["FAILED"] = True
        out["configs"][tag] = c
    return out


def roofline_block(acc, fp16, peak, nframes):
    """The `roofline` object of a leg from a profiled replay (profile_dominant).  fp32: the dominant kernel = the Winograd GEMMs (executed
    FLOP).  fp16 mode (since round 5): `frac` is over a FIXED set of layers -- every 3x3 conv that reads an fp16 map, whatever kernel it
    is routed to (tdnet_last_ms which = 4) -- so that routing a layer to a faster kernel can only raise it; the round-3/4 figure over the
    kernels selected by tile (which = 3: the LDS-DMA / 128x128-tile convs) is printed beside it."""
    dom_ms, dom_fl, dom_n = acc[3]
    ach = dom_fl / (dom_ms * 1e-3) / 1e12
    if not fp16:
        return {"bound": "mfma", "kernel": "k_gemm_dma / k_gemm_persistent<*,*,*,*,ROLE=1> (Winograd F(4x4) GEMMs, executed FLOP)",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "avg_launch_ms": round(dom_ms / dom_n, 4), "launches_per_frame": dom_n / nframes}
    fx_ms, fx_fl, fx_n = acc[4]
    fx = fx_fl / (fx_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "every 3x3 conv on an fp16 map (fixed layer set: k_conv_dma_h3n / _h3p / _h3 / _h<RH,3,..> / k_conv_igemm_h<..,3,true,..> / k_conv_igemm_h_group<..,3,1,true,..>), fp16 MFMA, fp32 accumulate",
            "achieved": round(fx, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(fx / peak, 4),
            "avg_launch_ms": round(fx_ms / fx_n, 4), "launches_per_frame": fx_n / nframes,
            "frac_round4_kernel_set": round(ach / peak, 4), "round4_kernel_set": "LDS-DMA + 128x128-tile 3x3 convs only (selected by kernel, %g launches per frame)" % (dom_n / nframes),
            "note": "per-launch durations (HIP events on the launch's own stream)"}


def main():
    t_proc0 = time.perf_counter()
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args, argv)                                        # does not return

    import numpy as np
    import torch
    import torch.distributed as dist
    from tdnet_amd import arch, parallel, weights

    H, W = (int(v) for v in args.size.lower().split("x"))
    if args.dry_run:
        H, W = 33, 65
    rank, local_rank, world = parallel.init_distributed("gloo" if (args.dry_run or args.share_gpu) else None)
    if args.share_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (there is no CPU path; --dry-run checks the launch plumbing only)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # N > 1: each rank next to its GPU's NUMA node, on its own cores, with an explicit intra-op thread count (the launcher exports
    # OMP_NUM_THREADS=1); N = 1 keeps the whole host (the CPU-oracle leg uses it)
    allowed0 = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    aff = None
    if world > 1:
        # ranks of THIS node share its cores: LOCAL_WORLD_SIZE (torch.distributed.run exports it), not the global world size
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        aff = parallel.pin_rank(local_rank, local_world, device_indices=[0] * local_world if (args.share_gpu or args.dry_run) else None)
    ones = parallel.allreduce_sum(torch.ones(1, dtype=torch.float64, device=dev))
    world_seen = int(round(ones.item()))
    if world_seen != args.gpus:
        raise SystemExit("bench.py: all-reduce saw %d ranks, expected %d" % (world_seen, args.gpus))
    backend = dist.get_backend() if dist.is_initialized() else "none"

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    kopts = {"winograd": args.winograd, "pipeline": args.conv_pipeline, "attention": args.attention, "fusion": args.fusion, "overlap": args.overlap, "gemm_persistent": args.gemm_persistent,
             "precision": 1 if args.precision == "fp16" else 2 if args.precision == "bf16x3" else None}
    kopts = {k: v for k, v in kopts.items() if v is not None}
    if args.backbone is None:
        args.backbone = "resnet101" if args.model == "psp" else "resnet18"
    spec = arch.model_spec(args.model, 19, args.backbone)
    P = spec.path_num
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = weights.synth_state_dict(spec, h, w, 0) if rank == 0 else None
    sync(); parallel.barrier()
    t0 = time.perf_counter()
    sd = parallel.broadcast_state_dict(spec, h, w, sd, dev)                       # the one collective on the data path
    sync(); parallel.barrier()
    bcast_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else 0.0
    nparam = sum(int(np.asarray(v).size) for v in sd.values())
    if args.perturb_rank == rank:                                                 # TEST ONLY: this rank now holds different weights
        k0 = sorted(k for k in sd if k.endswith("conv1.weight"))[0]
        sd = dict(sd)
        sd[k0] = np.asarray(sd[k0], np.float32) * np.float32(1.001)

    def make_model(opts, load=True):
        from tdnet_amd.model import pspnet, td2_psp50, td4_psp18
        if args.model == "psp":                                                   # the reference's comparison model (test.py:34-38)
            m = pspnet.pspnet(nclass=19, model_path=None, backbone=args.backbone, kernel_opts=opts)
        else:
            cls = td4_psp18.td4_psp18 if args.model == "td4" else td2_psp50.td2_psp50
            m = cls(nclass=19, path_num=P, model_path=None, backbone=args.backbone, kernel_opts=opts)
        m = m.eval().to(dev)
        if load:
            m.load_state_dict(sd)
        return m

    C = 1 if args.dry_run else max(1, args.clips_per_gpu)
    if args.mode == "frame-pipelined" and not args.dry_run:
        C = 2                                                                     # two handles, ONE clip (set up below)
    NF = 8
    if args.dry_run:
        models, streams, clips = [None], [None], [[torch.zeros(1) for _ in range(NF)]]
    else:
        # extra clips / the second lane of a frame-pipelined clip: own handle (workspace + FIFO + streams) on clip 0's WEIGHT BLOCK
        # (include/tdnet.h tdnet_create_shared) -- one copy of the packed weights per GPU however many streams it serves
        models = [make_model(kopts)]
        models[0].ensure_engine(H, W, dev)
        for _ in range(C - 1):
            models.append(make_model(kopts, load=False).share_weights_with(models[0]))
        # two lanes whatever C (model/_base.py _for_each_sample): even clips on the current stream, odd ones on ONE more stream that is
        # checked to sit on another hardware queue; a third concurrently busy stream costs more than it fills
        from tdnet_amd import _capi
        from tdnet_amd.model._base import _TDNetBase
        cur_s = torch.cuda.current_stream(dev)
        side_s = _TDNetBase._stream_beside([cur_s], dev, _capi.lib()) if C > 1 else None
        streams = [cur_s if c % 2 == 0 else side_s for c in range(C)]
        # one clip per handle (different seeds), pre-staged on the device; frames cycle, pos_id keeps counting
        clips = [[torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100 + rank + 1000 * c)] for c in range(C)]
    model, clip = models[0], clips[0]
    state = {"t": 0}

    pp = None
    if args.mode == "path-parallel" and not args.dry_run:
        pp = parallel.PathParallelStream(model, P, rank=rank, world=world, device=dev, frame_size=(H, W))
        clip = clips[0] = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, NF, seed=100)]   # the SAME stream on every rank
        C = 1

    fpl, fpl_same = None, None
    if args.mode == "frame-pipelined" and not args.dry_run:
        fpl = parallel.FramePipelinedStream(models[:2], P, dev, (H, W))
        C = 1
        # the line says "bit-identical outputs": hold the two-lane path to ONE handle on a replay from an empty FIFO, in this run
        with torch.no_grad():
            nchk_ = P + 4
            one = [models[0](clip[t % NF], pos_id=t % P).clone() for t in range(nchk_)]
            models[0].reset()
            two = fpl.process([clip[t % NF] for t in range(nchk_)])
            fpl_same = all(torch.equal(x, y) for x, y in zip(one, two))
            del one, two
            fpl.reset()
    gclip = {"g": None}
    if args.graph and (C != 1 or fpl is not None or pp is not None or args.dry_run):
        raise SystemExit("bench.py --graph: one clip per GPU in the default mode only")
    FR = C * (2 if fpl is not None else 1) * (P if args.graph else 1)   # frames per step and rank (--graph: a step = one cycle of P frames)

    def step(ms=None, n_clips=None):
        ms = models if ms is None else ms
        t = state["t"]
        if fpl is not None and ms is models:                          # one round: frames t, t + 1 of the clip, no join between rounds
            out = fpl.process([clip[t % NF], clip[(t + 1) % NF]], first_frame=t, join=False)
            state["t"] = t + 2
            return out[0]
        if gclip["g"] is not None and ms is models:                   # --graph: one replay = P frames (t is a multiple of P here)
            out = gclip["g"].replay([clip[(t + j) % NF] for j in range(P)])
            state["t"] = t + P
            return out[0]
        if args.dry_run:
            clip[t % NF].add_(1.0)
            state["t"] = t + 1
            return None
        if pp is not None:                                            # one round: frames t .. t + world - 1
            # PathParallelStream.process keeps pos_id = t mod P only if rounds start at multiples of lcm(world, P): NF = 8 does
            out = pp.process([clip[(t + j) % NF] for j in range(world)], first_frame=t)
            state["t"] = t + world
            return out
        out = None
        for c in range(len(ms) if n_clips is None else n_clips):
            with torch.cuda.stream(streams[c]):
                o = ms[c](clips[c][t % NF], pos_id=t % P)
            out = o if out is None else out
        state["t"] = t + 1
        return out

    def timed(nsteps, ms=None):
        sync(); parallel.barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(ms)
        sync(); parallel.barrier()
        return time.perf_counter() - t0

    # EXACTLY --warmup untimed steps when that fills the FIFO (>= P frames: the first steady-state frame is frame P - 1 .. P); fewer are raised to P and
    # the line says so (warmup = used, warmup_requested = asked)
    nwarm = max(args.warmup, P)
    with torch.no_grad():
        for _ in range(nwarm):
            step()
        if args.graph:                                                # capture at the start of a cycle, FIFO full; then warm the replay path as well
            from tdnet_amd.graph import GraphedClip
            while state["t"] % P or state["t"] < P:
                step()
            sync()
            gclip["g"] = GraphedClip(model, H, W, dev)
            step()
        sync()
        init_s = time.perf_counter() - t_proc0                        # process start -> first steady frame done (import, weights, handle, warm-up)
        # host cost of ENQUEUEING a frame (all of its launches, events and stream waits), measured on an empty queue so that the host
        # never waits for the device: when this approaches ms_per_step the rank is CPU-bound, which an 8-rank node can be with 8
        # launcher processes on one socket
        nh = 0 if args.pmc_child else 4                               # (a counter pass counts bytes per frame: no extra frames there)
        parallel.barrier()
        th0 = time.perf_counter()
        for _ in range(nh):
            step()
        host_us = (time.perf_counter() - th0) / max(nh * (P if args.graph else 1), 1) * 1e6
        sync()
        # N > 1: the SAME process's 1-rank figure, in the same invocation -- rank 0 runs the K steps alone while the others wait at a barrier -- so
        # that the line carries its own scaling efficiency (N-rank per-GPU frames/s / this) instead of leaning on another run of another box
        solo_dt = None
        if world > 1 and pp is None and not args.pmc_child:
            parallel.barrier()
            if rank == 0:
                sync()
                ts0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                sync()
                solo_dt = time.perf_counter() - ts0
            parallel.barrier()
        dt = timed(args.steps)
        # `value` stays the contract's EXACTLY K steps; a window under 0.5 s (K = 20 at 275 frames/s is 73 ms) is re-measured over as
        # many frames as fill 0.5 s -- the same number on every rank (derived from the all-reduced K-step time) -- and printed beside it
        sustained = None
        if not args.pmc_child and not args.dry_run and not args.quick and args.steps > 0:
            dt_all = parallel.allreduce_max(torch.tensor([dt], dtype=torch.float64, device=dev)).item()
            if dt_all < 0.5:
                n_long = int(min(20000, max(args.steps + 1, round(0.6 * args.steps / max(dt_all, 1e-6)))))
                dt_long = timed(n_long)
                sustained = (n_long, parallel.allreduce_max(torch.tensor([dt_long], dtype=torch.float64, device=dev)).item())
    if args.pmc_child:                                                # counter pass under rocprofv3: the steps above are all it needs
        return 0
    if args.graph:                                                    # everything below (rank check, profiled replay, parity) runs the eager frame calls again:
        gclip["g"] = None                                             # the ring is where the eager loop would have left it (t is a multiple of P)
    tmax = parallel.allreduce_max(torch.tensor([dt], dtype=torch.float64, device=dev)).item()
    per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
    per_rank[rank] = FR * args.steps / dt
    per_rank = parallel.allreduce_sum(per_rank).tolist()
    init_rank = torch.zeros(world, dtype=torch.float64, device=dev)
    init_rank[rank] = init_s
    init_rank = parallel.allreduce_sum(init_rank).tolist()
    host_rank = torch.zeros(world, dtype=torch.float64, device=dev)
    host_rank[rank] = host_us
    host_rank = parallel.allreduce_sum(host_rank).tolist()
    aff_rank = None
    if aff is not None:
        aff_rank = parallel.gather_strings("node %s cpus %s (%d), %d threads%s" % (aff.get("numa_node"), aff.get("cpus"), aff.get("n_cpus", 0), aff.get("omp_num_threads", 0),
                                                                               "" if aff.get("pinned") else " NOT PINNED: " + str(aff.get("why", ""))), world, dev)
    fps = world * FR * args.steps / tmax
    hwq_rank = parallel.gather_strings(__import__("tdnet_amd").hw_queue_note(), world, dev) if world > 1 else None

    mname = ("psp%s" if args.model == "psp" else args.model + "-psp%s") % args.backbone[6:]
    res = {"metric": "frames/sec (%s, %dx%d, full-resolution logits)" % (mname, H, W),
           "value": None if args.dry_run else round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": nwarm, "warmup_requested": args.warmup,
           "ms_per_step": round(1e3 * tmax / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if args.precision == "fp32" else "f32 via bf16x3 split, fp32 accumulate (GEMM operands as three bf16 parts, six products on the bf16 MFMA)" if args.precision == "bf16x3"
                    else "f16 (fp16 MFMA convs + attention, fp16 activation maps in the backbone, fp32 accumulate / softmax / LayerNorm)",
           "data": "synthetic",
           "world_size_seen": world_seen, "backend": backend, "rccl_bcast_ms": round(bcast_ms, 3),
           "bcast_bytes": 4 * nparam if world > 1 else 0, "per_rank_fps": [round(v, 3) for v in per_rank],
           "per_rank_fps_min_max_spread": [round(min(per_rank), 3), round(max(per_rank), 3), round((max(per_rank) - min(per_rank)) / max(max(per_rank), 1e-9), 4)],
           "init_s_per_rank": [round(v, 2) for v in init_rank],
           "host_launch_us_per_frame": [round(v, 1) for v in host_rank],
           "cpu_affinity": aff_rank if aff_rank is not None else "not pinned (N = 1: the whole host, %d CPUs allowed)" % (len(allowed0) if allowed0 else os.cpu_count() or 1),
           "omp_num_threads": (aff or {}).get("omp_num_threads", torch.get_num_threads()),
           "hw_queues": __import__("tdnet_amd").hw_queue_note(),
           "config": {"workload": "%s, %dx%d Cityscapes-shaped synthetic stream, %d-frame feature cache, %d clip%s per GPU"
                                  % (mname, H, W, spec.fifo, C, "" if C == 1 else "s (concurrent HIP streams)"),
                      "parallelism": ("clip-parallel x%d, RCCL weight broadcast only; each clip with TWO FRAMES IN FLIGHT (two handles sharing one weight block on two "
                                      "HIP streams, cache entries handed over between encode and propagate; outputs held bit for bit to one handle in this run: "
                                      "see two_lanes_bit_identical_to_one_handle)" % world) if fpl is not None else
                                     ("clip-parallel x%d, RCCL weight broadcast only" % world) if pp is None else
                                     ("path-parallel x%d: one stream, one all-gather of %d cache entries per round" % (world, world)),
                      "target_fps_per_gpu": 30}}
    if args.graph:
        res["config"]["step"] = "--graph: a step = ONE hipGraph replay = one pos_id cycle of %d frames (frame calls captured on a stream, tdnet_amd/graph.py); value counts frames" % P
    if world > 1:
        ver = None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            ver = None
        res["collective"] = {"backend": backend, "rccl_version": ver, "bcast_ms": round(bcast_ms, 3), "bcast_bytes": 4 * nparam,
                             "bcast_GBps": round(4 * nparam / max(bcast_ms, 1e-6) / 1e6, 2),
                             "what": "ONE broadcast of the flat weight blob at start (timed between two barriers, so it includes the slowest rank's readiness); "
                                     "end-of-run all-reduces of timing / digests / confusion matrix; no per-frame collective",
                             "hw_queues_per_rank": hwq_rank}
        if solo_dt is not None or rank != 0:
            solo_fps = None if solo_dt is None else FR * args.steps / solo_dt
            res["scaling_check"] = {"solo_fps_rank0": None if (solo_fps is None or args.dry_run) else round(solo_fps, 3),
                                    "per_gpu_fps": None if args.dry_run else round(fps / world, 3),
                                    "scaling_efficiency": None if (solo_fps is None or args.dry_run) else round(fps / world / solo_fps, 4),
                                    "what": "rank 0 alone ran the same K steps in this invocation (the other ranks waiting at a barrier) right before the N-rank timed "
                                            "loop: efficiency = N-rank frames/s per GPU / that.  No curve has been measured by the builder (no multi-GPU box)"}
    if sustained is not None:
        res["sustained"] = {"steps_timed": sustained[0], "seconds": round(sustained[1], 4),
                            "value": round(world * FR * sustained[0] / sustained[1], 3),
                            "unit": "frames/s", "ms_per_step": round(1e3 * sustained[1] / sustained[0], 4),
                            "note": "the same timed loop (barrier + synchronize on both sides, max over ranks) over >= 0.5 s, because --steps %d "
                                    "is a %.0f-ms window; `value` above is exactly --steps frames" % (args.steps, 1e3 * tmax)}
    if fpl is not None:
        res["two_lanes_bit_identical_to_one_handle"] = bool(fpl_same)
        if not fpl_same:
            res["FAILED"] = "frame-pipelined outputs differ from the single handle's"
    if pp is not None:
        res["scaling"] = "strong"
    if args.share_gpu:
        res["shared_gpu"] = True
        res["value"] = None
        res["metric"] = "SHARED GPU (N ranks on one device over gloo: code-path check, not a measurement): " + res["metric"]
    if args.dry_run:
        res["dry_run"] = True
        res["metric"] = "DRY RUN (launch plumbing only, no model): " + res["metric"]

    exit_code = 5 if (fpl is not None and not fpl_same) else 0
    # ---- N > 1: the line must be able to FAIL.  Every rank replays RANK 0's clip for P + 2 frames and contributes a 64-bit digest of each
    # frame's logits bits plus its label histogram; MIN and MAX all-reduces must coincide (every rank bit-identical to rank 0, whose
    # output is held to the CPU oracle below), and the 19x19 confusion matrices (each rank's labels vs the oracle's) are
    # summed over the ranks into the line.  --dry-run does the same over the broadcast weight bytes (there is no model on the CPU).
    ref_labels = None
    if world > 1 and pp is None:
        nchk = P + 2
        my_labels = []
        if args.dry_run:
            blob = np.concatenate([np.asarray(sd[k], np.float32).reshape(-1) for k in sorted(sd)])
            vec = logits_digest(torch.from_numpy(blob), torch)
        else:
            clip0 = clip if rank == 0 else [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, nchk, seed=100)]
            vec = []
            fpl.reset() if fpl is not None else model.reset()            # (two frames in flight: both lanes' FIFOs)
            with torch.no_grad():
                for t in range(nchk):
                    out = model(clip0[t % len(clip0)], pos_id=t % P)
                    lab = out[0].argmax(0)
                    my_labels.append(lab.to(torch.uint8))
                    vec += logits_digest(out, torch) + torch.bincount(lab.flatten(), minlength=19)[:19].tolist()
            model.reset()
            if fpl is not None:
                fpl.reset()
        v = torch.tensor(vec, dtype=torch.int64, device=dev)
        vmin, vmax = v.clone(), v.clone()
        dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        agree = bool(torch.equal(vmin, vmax))
        res["rank_check"] = {"ranks_agree": agree, "frames": 0 if args.dry_run else nchk, "digest_words": len(vec),
                             "what": ("digest of the broadcast weight bytes (dry run: no model)" if args.dry_run else
                                      "every rank replays rank 0's clip: per frame two 64-bit digests of the logits bits + the label "
                                      "histogram; all_reduce(MIN) == all_reduce(MAX)")}
        if not agree:
            res["rank_check"]["FAILED"] = True
            res["rank_check"]["words_differing"] = int((vmin != vmax).sum().item())
            exit_code = 4
        if not args.dry_run and not args.no_cpu_baseline:
            # rank 0 evaluates the oracle on those frames (bounded: P + 2 frames), everyone receives its labels, every rank's confusion
            # matrix against them is summed: the end-of-run all-reduce of SURVEY 8e
            lab_ref = torch.zeros((nchk, H, W), dtype=torch.uint8, device=dev)
            if rank == 0:
                from oracle import tdnet_ref                                      # checker only
                if allowed0 is not None:
                    os.sched_setaffinity(0, allowed0)                             # the timed region is over: the oracle gets the whole host back
                tdnet_ref.tune_threads()
                oracle = tdnet_ref.TDNetRef(spec, sd)
                keep = []
                par0, _, _ = parity_sample(model, oracle, clip, P, nchk, 0, args.precision != "fp16", torch, np, tdnet_ref, keep_labels=keep)
                model.reset()
                res["parity"] = par0
                if par0.get("FAILED"):
                    exit_code = 3
                lab_ref.copy_(torch.from_numpy(np.stack(keep)))
            dist.broadcast(lab_ref, src=0)
            mine = torch.stack(my_labels).to(torch.int64)
            cm = torch.bincount((19 * lab_ref.to(torch.int64) + mine).flatten(), minlength=361)[:361].reshape(19, 19)
            cm = parallel.allreduce_sum(cm)
            cmn = cm.cpu().numpy().astype(np.float64)
            iu = np.diag(cmn) / np.maximum(1, cmn.sum(1) + cmn.sum(0) - np.diag(cmn))
            res["rank_check"]["confusion_matrix_sum_over_ranks"] = cm.cpu().tolist()
            res["rank_check"]["miou_vs_cpu_all_ranks"] = round(float(iu[cmn.sum(1) > 0].mean()), 6)
            res["rank_check"]["pixels_all_ranks"] = int(cmn.sum())
        code = torch.tensor([exit_code], dtype=torch.int64, device=dev)
        dist.all_reduce(code, op=dist.ReduceOp.MAX)                            # every rank leaves with the same verdict
        exit_code = int(code.item())
    if rank == 0 and not args.dry_run:
        eng = model.engine
        opts = eng.opts()
        gflop = eng.flops_per_frame() / 1e9
        peak = PEAK_FP16_MFMA_TFLOPS if opts["precision"] == 1 else PEAK_BF16X3_TFLOPS if opts["precision"] == 2 else PEAK_FP32_MFMA_TFLOPS
        res["config"]["kernel_opts"] = opts
        # ---- roofline of the dominant kernel: profiled replay (HIP events around every launch, same stream) ----------
        nprof = 2 * P
        if world > 1 and pp is None:
            # the rank check above left the FIFO empty (model.reset()): refill it, or the profiled frames would be warm-up frames
            # without the attention chain and the per-frame launch counts of the N-rank line would not be steady-state ones
            with torch.no_grad():
                for _ in range(P + 2):
                    step(n_clips=1)
            sync()
        # the reference's own protocol (test.py:46-59: a device synchronisation on both sides of every frame, frames i > 5 averaged),
        # clip 0 alone -- the like-for-like partner of BASELINE.md's published ms/frame -- beside the throughput `value`
        if pp is None and fpl is None:
            with torch.no_grad():
                res["latency_ms_synced"] = round(latency_synced(lambda: step(n_clips=1), sync), 4)
            res["latency_note"] = ("latency_ms_synced = Testing/test.py:46-59's loop (synchronise, start, forward, synchronise; frames i > 5 averaged), one "
                                   "clip, frames resident in HBM; `value` = K frames back to back between one pair of synchronisations (throughput)")
        wbytes, hbytes, nshare = eng.memory_bytes()
        res["memory"] = {"weights_bytes": wbytes, "handle_bytes": hbytes, "handles_on_this_weight_block": nshare,
                         "note": "a further stream on this GPU (batch sample, clip, pipeline lane) costs handle_bytes only: tdnet_create_shared"}
        acc = profile_dominant(eng, lambda: step(n_clips=1), nprof, sync, torch)   # the replay runs clip 0 alone: per-launch durations
        res["launches_per_frame"] = eng.last_launch_count()
        dom_ms, dom_fl, dom_n = acc[3]
        dom_regex = None
        if dom_n > 0 and dom_ms > 0:
            achieved = dom_fl / (dom_ms * 1e-3) / 1e12
            if opts["precision"] == 1:
                kname, dom_regex = ("k_conv_dma_h3<RH,..> / k_conv_dma_h3p<RH,..> (dedicated loader waves) / k_conv_dma_h3n<RH,..> (narrow tiles) / k_conv_dma_h<RH,3,..> (3x3 dilated convs on fp16 maps, fp16 MFMA fed by LDS-DMA, fp32 accumulate; td_conv_hd.h) + "
                                    "k_conv_igemm_h<..,3,true,..> where the register-staged kernel is kept"), r"k_conv_dma_h3[a-z]?<|k_conv_dma_h<\d, 3|k_conv_igemm_h<\d+, \d+, \d, \d, 3, true|k_conv_igemm_h_group<\d+, \d+, \d, \d, 3, 1, true"
            elif opts["winograd"]:
                f4 = opts["winograd"] >= 3
                if opts["precision"] == 2 and opts["gemm_persistent"]:
                    gk, dom_regex = ("k_gemm_b3m<1> (fp32 operands as three bf16 parts, six bf16-MFMA products, fp32 accumulate; td_gemm_b3.h) [+ k_gemm_dma where a GEMM is too small for it]",
                                     r"k_gemm_b3m?<|k_gemm_dma\(|k_gemm_persistent<\d+, \d+, \d+, \d+, 1>")
                elif opts["gemm_persistent"]:
                    gk, dom_regex = ("k_gemm_dma (LDS-DMA-fed, tdnet_opts.overlap bit 8) / k_gemm_persistent<*,*,*,*,ROLE=1>",
                                     r"k_gemm_dma\(|k_gemm_persistent<\d+, \d+, \d+, \d+, 1>")
                else:
                    gk, dom_regex = "k_conv_igemm<.,.,.,.,1> (batched)", r"k_conv_igemm<\d+, \d+, \d+, \d+, 1, false"
                kname = (gk + ": the %d batched GEMMs of the Winograd F(%s,3x3) convs of layers %s + head, fp32 MFMA; FLOP = executed "
                         "GEMM FLOP, %sx fewer than the direct conv's" % ((36, "4x4", "2-4", "4") if f4 else (16, "2x2", "3-4", "2.25")))
            else:
                kname, dom_regex = "k_conv_igemm<128,128,2,2,3> (3x3 dilated conv, fp32 MFMA)", r"k_conv_igemm<128, 128, 2, 2, 3, false"
            res["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(achieved / peak, 4), "traffic": None,
                               "avg_launch_ms": round(dom_ms / dom_n, 4), "launches_per_frame": dom_n / nprof,
                               "gflop_per_launch": round(dom_fl / dom_n / 1e9, 2)}
            if opts["precision"] == 1 and acc[4][2] > 0:                    # fp16 mode: the FIXED layer set is the headline fraction (roofline_block)
                res["roofline"].update(roofline_block(acc, True, peak, nprof))
                res["roofline"]["traffic"] = None
        # With the row-parity chains (tdnet_opts.overlap) two launches of the dominant kernel are in flight at a time, each progressing at
        # about half speed: `frac` (per-launch durations, the figure rocprofv3's kernel stats reproduce) then understates the kernel.
        # The same replay on a handle WITHOUT the chains gives the kernel's own rate, reported beside it.
        if "roofline" in res and opts.get("overlap", 0) & 1 and opts["winograd"] >= 3 and opts["precision"] != 1 and world == 1 and pp is None and not args.no_direct_line:
            ms_ = make_model(dict(kopts, overlap=0))
            st2 = {"t": 0}

            def step_serial():
                t = st2["t"]
                ms_(clip[t % NF], pos_id=t % P)
                st2["t"] = t + 1
            with torch.no_grad():
                for _ in range(P + 2):
                    step_serial()
            acc_s = profile_dominant(ms_.engine, step_serial, nprof, sync, torch)
            if acc_s[3][2] > 0 and acc_s[3][0] > 0:
                a_s = acc_s[3][1] / (acc_s[3][0] * 1e-3) / 1e12
                res["roofline"]["serial_launches"] = {"achieved": round(a_s, 2), "frac": round(a_s / peak, 4),
                                                      "avg_launch_ms": round(acc_s[3][0] / acc_s[3][2], 4), "launches_per_frame": acc_s[3][2] / nprof,
                                                      "note": "the same kernel on a handle with kernel_opts overlap=0 (one launch at a time, whole convs): "
                                                              "its own rate; in the default configuration two launches run concurrently and share the CUs"}
            ms_.engine.close()                                          # streams of a handle are hardware queues: release them now
            del ms_
        exec_gflop = (acc[0][1] + acc[1][1]) / nprof / 1e9           # conv/GEMM (executed: Winograd GEMM FLOP) + attention matmuls
        single_fps = FR * args.steps / tmax                           # this GPU's frames/s
        res["frame"] = {"algorithmic_gflop": round(gflop, 1), "algorithmic_tflops": round(gflop * single_fps / 1e3, 2),
                        "algorithmic_frac_of_roof": round(gflop * single_fps / 1e3 / peak, 4),
                        "executed_gflop": round(exec_gflop, 1), "executed_tflops": round(exec_gflop * single_fps / 1e3, 2),
                        "executed_frac_of_roof": round(exec_gflop * single_fps / 1e3 / peak, 4),
                        "note": "algorithmic = the reference's op list (SURVEY 8d); executed = what the kernels multiply (a Winograd "
                                "F(4x4) conv runs 1/4 of the direct conv's MACs), so only executed/peak is a roofline fraction"}
        res["breakdown_ms_per_frame"] = {"conv_gemm": round(acc[0][0] / nprof, 3), "attention": round(acc[1][0] / nprof, 3),
                                         "hbm_bound_tail": round(acc[2][0] / nprof, 3)}
        res["breakdown_tflops"] = {"conv_gemm": round(acc[0][1] / max(acc[0][0], 1e-9) / 1e9, 2),
                                   "attention": round(acc[1][1] / max(acc[1][0], 1e-9) / 1e9, 2)}

        # ---- the arithmetic-faithful configuration beside the headline: every conv direct (no Winograd) -------------------
        if world == 1 and opts["winograd"] and not args.no_direct_line and pp is None:
            md = make_model(dict(kopts, winograd=0))
            nd = max(8, args.steps // 3)
            with torch.no_grad():
                for _ in range(P + 2):
                    step([md])
                dtd = timed(nd, [md])
            dfps = nd / dtd
            res["frame"]["all_direct"] = {"value": round(dfps, 3), "unit": "frames/s", "ms_per_step": round(1e3 * dtd / nd, 4), "steps": nd,
                                          "frac_of_roof": round(gflop * dfps / 1e3 / peak, 4),
                                          "note": "same frame with every conv as a direct implicit GEMM (kernel_opts winograd=0): executed "
                                                  "FLOP = algorithmic FLOP"}
            md.engine.close()
            del md

        # ---- HBM traffic from live PMC passes (N = 1 only; each pass re-runs a short bench under rocprofv3) ----------------
        if world == 1 and dom_regex and not args.no_pmc:
            psteps, pwarm = 4, P + 2
            child = ["--pmc-child", "--steps", str(psteps), "--warmup", str(pwarm), "--no-cpu-baseline", "--no-pmc", "--model", args.model,
                     "--backbone", args.backbone, "--size", args.size, "--precision", args.precision, "--clips-per-gpu", "1"]
            for k_, v_ in (("--winograd", args.winograd), ("--conv-pipeline", args.conv_pipeline), ("--attention", args.attention), ("--fusion", args.fusion), ("--overlap", args.overlap), ("--gemm-persistent", args.gemm_persistent)):
                if v_ is not None:
                    child += [k_, str(v_)]
            per_launch, per_frame, detail = measure_traffic(child, dom_regex, psteps + pwarm)
            if "roofline" in res:
                res["roofline"]["traffic"] = None if per_launch is None else round(per_launch)
                res["roofline"].update(detail)
            if per_frame is not None:
                res["frame"]["hbm_bytes_all_kernels"] = round(per_frame)
        elif "roofline" in res:
            res["roofline"]["traffic_source"] = "not measured (--no-pmc or N > 1)"

        # ---- the steady state as one hipGraph per cycle, beside the eager loop (N = 1, one clip) ---------------------------------
        if world == 1 and pp is None and fpl is None and C == 1 and not args.quick:
            res["hipgraph"] = graph_compare(model, clip, P, NF, H, W, dev, sync, args.steps, torch)
            if not res["hipgraph"]["bit_identical_to_eager"]:
                res["hipgraph"]["FAILED"] = True
                exit_code = 5

        # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only) ------------------------------------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import tdnet_ref                                          # checker / baseline only
            cores = tdnet_ref.tune_threads()          # threads actually used (fastest of 8..128 on a probe conv)
            ref = (tdnet_ref.PSPNetRef if args.model == "psp" else tdnet_ref.TDNetRef)(spec, sd)
            nw, nsteady = P, max(1, args.cpu_frames)
            par, cpu_t, _ = parity_sample(model, ref, clip, P, nw, nsteady, args.precision != "fp16", torch, np, tdnet_ref)
            res["cpu_baseline"] = {"value": round(nsteady / cpu_t, 4), "unit": "frames/s", "cores": cores, "host_threads": os.cpu_count() or 1, "kind": "port",
                                   "sample": "%d steady-state frames of the same clip (after %d warm-up frames), oracle/tdnet_ref.py "
                                             "= the reference's op graph on torch-CPU %s with %d threads (host has %d)"
                                             % (nsteady, nw, torch.__version__, cores, os.cpu_count() or 1)}
            res["parity"] = par
            if par.get("FAILED"):
                exit_code = 3
            del ref

        # ---- the other single-GPU configs of BASELINE.json on the same line (default N = 1 run only) --------------------------
        default_workload = args.model == "td4" and args.backbone == "resnet18" and (H, W) == (1024, 2048) and args.precision == "fp32" and C == 1
        if world == 1 and pp is None and default_workload and not args.no_other_configs:
            ncpu = 0 if args.no_cpu_baseline else 5                       # >= 5 steady-state frames on the CPU
            # the main handle is done: release it (and its internal streams) before the legs create theirs -- with the idle handles
            # of the earlier legs alive the first leg measured 228 frames/s against 333 alone (HIP maps streams onto few hardware queues)
            for m_ in models:
                if m_.engine is not None:
                    m_.engine.close()
            res["other_configs"] = [
                # OPT-IN arithmetic (the headline stays exact fp32): the same workload with the large GEMMs as six bf16-MFMA products per fp32 product
                other_config_leg("configs[2] with tdnet_opts.precision = 2 (opt-in)", "td4", "resnet18", (1024, 2048), "bf16x3", 40, min(ncpu, 3), dev, sync,
                                 "the headline workload; resnet.py:25-59 / transformer.py:126-139 are fp32 throughout: this mode keeps fp32 storage, fp32 accumulation and "
                                 "fp32-accurate products (2^-26), and is held to the fp32 parity gate", graph=True),
                other_config_leg("configs[1]", "td2", "resnet18", (1024, 2048), "fp32", 40, ncpu, dev, sync,
                                 "td2_psp50(backbone='resnet18', path_num=2), Testing/model/pspnet/td2_psp50.py:52-58"),
                other_config_leg("configs[4]", "td2", "resnet34", (720, 960), "fp16", 40, ncpu, dev, sync,
                                 "td2-bise34 does not exist in the reference (SURVEY 0): td2_psp50(backbone='resnet34') is its stand-in; "
                                 "fp16 MFMA with fp32 accumulation, parity reported against the fp32 CPU path", pipelined=True, graph=True),
                # the configurations the reference SHIPS and publishes numbers for, at its native 769x1537 (Testing/test.py:22-38, TEST_README.md:31-33)
                other_config_leg("native td4-psp18", "td4", "resnet18", (769, 1537), "fp32", 40, ncpu, dev, sync,
                                 "test.py:24-26 `--model td4-psp18` at 769x1537; TEST_README.md:33 publishes 85 ms/frame (Titan Xp)", published_ms=85.0),
                other_config_leg("native td2-psp50", "td2", "resnet50", (769, 1537), "fp32", 30, ncpu, dev, sync,
                                 "test.py:28-32 `--model td2-psp50` at 769x1537; TEST_README.md:32 publishes 180 ms/frame (Titan Xp)", published_ms=180.0),
                other_config_leg("native psp101", "psp", "resnet101", (769, 1537), "fp32", 20, min(ncpu, 3), dev, sync,
                                 "test.py:34-38 `--model psp101` at 769x1537; TEST_README.md:31 publishes 360 ms/frame (Titan Xp); CPU leg: 3 frames "
                                 "(the stateless ResNet-101 frame is the most expensive on the host)", published_ms=360.0)]
            if any(l.get("parity", {}).get("FAILED") for l in res["other_configs"]):
                exit_code = 3
            if not args.no_numerics_stress and not args.no_cpu_baseline:
                res["numerics_stress"] = numerics_stress(dev, torch, np)
                if res["numerics_stress"].get("FAILED"):
                    exit_code = 3
    if rank == 0:
        print(json.dumps(res), flush=True)
    parallel.barrier()
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
