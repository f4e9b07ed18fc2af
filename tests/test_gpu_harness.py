"""N2 on the GPU: PNG frames -> cityscapesLoader -> checkpoint file -> the reference's frame loop (tdnet_amd/test.py,
mirror of Testing/test.py) -> colour PNGs, checked against the CPU oracle fed with the same loader output."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import tdnet_ref
from tdnet_amd import arch, weights
from tdnet_amd.dataloader import cityscapesLoader

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_cli_end_to_end(tmp_path):
    from PIL import Image
    H, W, T = 129, 257, 7
    frames_dir = tmp_path / "data" / "vid1"
    frames_dir.mkdir(parents=True)
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, (64, 128, 3), dtype=np.uint8)
    for t in range(T):
        img = np.roll(base, 3 * t, axis=1)
        Image.fromarray(img).resize((512, 256), Image.BILINEAR).save(frames_dir / ("frame_%06d_leftImg8bit.png" % t))
    spec = arch.model_spec("td4", 19, "resnet18")
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = weights.synth_state_dict(spec, h, w, 0)
    ckpt = tmp_path / "td4-psp18.pkl"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(ckpt))
    out_dir = tmp_path / "output"
    out_dir.mkdir()
    r = subprocess.run([sys.executable, "-m", "tdnet_amd.test", "--model", "td4-psp18", "--img_path", str(tmp_path / "data"),
                        "--output_path", str(out_dir), "--_td4_psp18_path", str(ckpt), "--in_size", "%dx%d" % (H, W)],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Average  RunningTime/Latency" in r.stdout and " Frame  7" in r.stdout
    # the oracle on the loader's tensors must produce the same quarter-resolution colour maps
    ld = cityscapesLoader(img_path=str(tmp_path / "data"), in_size=(H, W))
    ld.load_frames()
    ref = tdnet_ref.TDNetRef(spec, sd)
    mism = 0
    for t, (img, name, folder, size) in enumerate(ld.data):
        exp = ref.forward(img, t % 4)[0].argmax(0).numpy().astype(np.int8)
        oh, ow = size[1] // 4, size[0] // 4
        ys = np.minimum((np.arange(oh) * (H / oh)).astype(np.int64), H - 1)
        xs = np.minimum((np.arange(ow) * (W / ow)).astype(np.int64), W - 1)
        exp_rgb = ld.decode_segmap(exp[ys][:, xs]).astype(np.uint8)
        got = np.asarray(Image.open(out_dir / folder / name).convert("RGB"))
        assert got.shape == exp_rgb.shape == (oh, ow, 3)
        mism += int((got != exp_rgb).any(axis=2).sum())
    assert mism <= 0.002 * T * oh * ow, mism          # only numerical-tie pixels may differ
    # the throughput loop (--prefetch: DevicePrefetcher + forward_labels + LabelDownloader) must write the very same PNGs
    out2 = tmp_path / "output_prefetch"
    out2.mkdir()
    r = subprocess.run([sys.executable, "-m", "tdnet_amd.test", "--model", "td4-psp18", "--img_path", str(tmp_path / "data"),
                        "--output_path", str(out2), "--_td4_psp18_path", str(ckpt), "--in_size", "%dx%d" % (H, W), "--prefetch"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "%d frames, prefetched upload" % T in r.stdout, r.stdout
    for t, (img, name, folder, size) in enumerate(ld.data):
        a_, b_ = np.asarray(Image.open(out_dir / folder / name)), np.asarray(Image.open(out2 / folder / name))
        assert np.array_equal(a_, b_), (t, name)


def test_loaded_library_was_built_from_the_shipped_sources():
    """The .so travels prebuilt to the GPU box: its stamp (tdnet_version()) must equal the hash of the csrc/ + include/tdnet.h that
    travelled with it, or this run would be testing some other build's kernels."""
    from tdnet_amd import _capi, build as b
    ver = _capi.lib().tdnet_version().decode()
    assert ver.endswith("tdnet-src-hash:" + b.source_hash()), (ver, b.source_hash())


def test_bench_two_ranks_sharing_the_gpu():
    """The N-rank code path of bench.py with the real kernels: `--gpus 2 --share-gpu` launches two ranks itself, both on cuda:0, the
    219 MB weight blob crosses the process boundary (gloo), each rank builds its model from the broadcast copy and serves its own
    clip, timing is reduced over ranks.  (RCCL itself needs two devices: the driver's scaling run is the measurement.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "6", "--warmup", "6",
                        "--size", "257x513"], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2 and line["shared_gpu"] is True and line["value"] is None
    assert len(line["per_rank_fps"]) == 2 and min(line["per_rank_fps"]) > 0 and line["bcast_bytes"] > 2e8
    # the line verifies itself: both ranks replayed rank 0's clip and produced bit-identical logits; rank 0's are held to the oracle;
    # the confusion matrices of both ranks against the oracle's labels were all-reduced into the line
    rc = line["rank_check"]
    assert rc["ranks_agree"] is True and rc["frames"] == 6 and rc["miou_vs_cpu_all_ranks"] >= 0.9995
    assert rc["pixels_all_ranks"] == 2 * 6 * 257 * 513 and line["parity"]["flips_outside_tie_band"] == 0
    assert len(line["init_s_per_rank"]) == 2 and len(line["host_launch_us_per_frame"]) == 2 and len(line["cpu_affinity"]) == 2


def test_bench_four_ranks_one_perturbed_rank_fails():
    """FOUR ranks on the one GPU (per-process queues / VRAM of four handles side by side: VERDICT r3 item 3d), the LAST one with its
    weights perturbed after the broadcast (--perturb-rank): real kernels, different logits -> the digests disagree, the line says so
    and the run exits non-zero.  The line carries every rank's CPU placement and host launch cost."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--share-gpu", "--steps", "4", "--warmup", "6",
                        "--size", "129x257", "--perturb-rank", "3", "--no-cpu-baseline"], env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode != 0, out[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["rank_check"]["ranks_agree"] is False and line["rank_check"]["FAILED"] is True
    assert line["n_gpus"] == 4 and line["world_size_seen"] == 4 and len(line["per_rank_fps"]) == 4 and min(line["per_rank_fps"]) > 0
    assert len(line["cpu_affinity"]) == 4 and len(line["host_launch_us_per_frame"]) == 4 and min(line["host_launch_us_per_frame"]) > 0


def test_a_handle_created_behind_an_rccl_communicator_runs_at_full_rate():
    """The order of every multi-GPU run: init_process_group("nccl") (torch creates its stream pools, RCCL its queues), THEN the model.
    With HIP's default of 4 hardware queues per priority class such a handle ran at 0.67x (185 instead of 275 frames/s: the process owns
    more queues than stay resident and the frame's three streams are time-sliced); the package sets GPU_MAX_HW_QUEUES=2 at import
    (tdnet_amd/__init__.py).  World size 1 over nccl on the box's one GPU, tools/rccl_streams_probe.py: the rate behind the communicator
    must be within 8 % of the rate without one; the old default is run as well and printed, not asserted (it is the platform's behaviour)."""
    import re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GPU_MAX_HW_QUEUES")}

    def rate(mode, **env):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_streams_probe.py"), mode], env=dict(base, **env), cwd=root,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode(errors="replace")
        m = re.search(r": ([0-9.]+) frames/s", out)
        assert r.returncode == 0 and m, out[-2000:]
        return float(m.group(1))
    plain, behind = rate("none"), rate("nccl_first")
    old = rate("nccl_first", GPU_MAX_HW_QUEUES="4")
    print("td4-psp18 1024x2048: %.1f frames/s alone, %.1f behind an RCCL communicator (package default GPU_MAX_HW_QUEUES=2), %.1f with HIP's default of 4"
          % (plain, behind, old))
    assert behind >= 0.92 * plain, (plain, behind, old)


def test_idle_handles_do_not_slow_a_busy_one():
    """tools/idle_handle_probe.py: the C2 workload beside 0 / 1 / 2 / 3 idle td4 handles.  HIP reuses hardware queues once its pool is full;
    with an odd number of idle handles alive the busy handle's second row-parity chain used to land on the CALLER's queue and the frame ran
    at 0.63x (335 / 212 / 335 / 212 frames/s).  The first frame now checks the pair with two spin kernels and replaces the internal stream
    (td_frame.h place_chain_stream).  Run under HIP's default pool size (4), where the collision occurs.
    What is ASSERTED is the check's own measurement, which is deterministic: every handle's final spin pair ran side by side (<= 64 us; two
    40-us kernels take 40-51 us on two queues and 80-85 us on one: profiles/r04k_*, r05n_*).  The throughputs -- with the check and, for the
    record, without it (TDNET_NO_QUEUE_CHECK=1) -- are printed, not asserted: a frames/s figure on a box this test does not own moved by
    more than the effect in one of ~10 suite runs of round 5 (and this test then retried; it no longer does)."""
    import re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}

    def run(**extra):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "idle_handle_probe.py")], env=dict(env, GPU_MAX_HW_QUEUES="4", TDNET_QUIET="1", **extra),
                           cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode(errors="replace")
        v = [float(x) for x in re.findall(r"alive: ([0-9.]+) frames/s", out)]
        assert r.returncode == 0 and len(v) == 4, out[-2000:]
        return v, out
    got, out = run(TDNET_QUEUE_CHECK_VERBOSE="1")
    old, _ = run(TDNET_NO_QUEUE_CHECK="1")
    print("busy handle beside 0..3 idle ones, frames/s: %s with the queue check, %s without" % (got, old))
    # the probe's output, in order: spin-pair lines of a handle's check(s) (one per attempt; a shared pair is followed by the replacement's), then
    # that handle's "alive" line (the busy td2 handle) or nothing (an idle td4 handle's own first frame).  The LAST pair before each "alive" line
    # is the busy handle's final placement.
    finals, last = [], None
    for l in out.splitlines():
        m = re.search(r"queue check: spin pair ([0-9.]+) us", l)
        if m:
            last = float(m.group(1))
        elif "alive:" in l:
            finals.append(last)
            last = None
    print("final spin pair of the busy handle beside 0..3 idle ones: %s us" % finals)
    assert len(finals) == 4 and all(f is not None and f <= 64.0 for f in finals), (finals, got)


def test_bench_line_carries_measured_hbm_traffic_of_the_dominant_kernel():
    """`roofline.traffic` comes from live rocprofv3 counter passes matched to the dominant kernel BY NAME: when the default GEMM kernel
    changed (k_gemm_persistent -> k_gemm_dma) the pattern went stale and the field silently became null.  A quarter-size frame, both legs:
    the default fp32 path and the fp16 mode; the traffic must be there, positive, and of the order of the launch's algorithmic bytes."""
    import json, shutil, subprocess, sys
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for extra in ([], ["--precision", "fp16"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "6", "--size", "513x1025", "--no-cpu-baseline",
                            "--no-direct-line", "--no-other-configs"] + extra, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        out = r.stdout.decode(errors="replace")
        assert r.returncode == 0, out[-3000:]
        line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        roof = line["roofline"]
        assert roof["traffic"] is not None and roof["traffic"] > 0, (extra, roof)
        assert line["frame"].get("hbm_bytes_all_kernels", 0) > roof["traffic"], (extra, line["frame"])
