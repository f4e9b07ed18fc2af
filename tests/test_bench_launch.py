"""bench.py's launch contract (CPU, gloo): `python bench.py --gpus N` WITHOUT a launcher starts the N ranks itself and proves
they met (world_size_seen from an all-reduce), the launcher form `python -m torch.distributed.run ... bench.py --gpus N` works
the same, a WORLD_SIZE that disagrees with --gpus is an error, and the N = 1 line keeps its shape.  --dry-run swaps the model
for a no-op (there is no CPU path for the model) but runs the real launch / rendezvous / weight-broadcast / barrier / timing-
reduction code; its `value` is null so it can never be mistaken for a measurement."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "2"
    return e


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_self_spawn_two_ranks_over_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2 and line["backend"] == "gloo"
    assert line["dry_run"] is True and line["value"] is None          # plumbing only: never a measurement
    assert len(line["per_rank_fps"]) == 2 and all(v > 0 for v in line["per_rank_fps"])
    assert line["rccl_bcast_ms"] > 0 and line["bcast_bytes"] > 1e6      # the weight blob really crossed the process boundary
    assert line["steps"] == 2 and line["scaling"] == "weak"
    assert sum(1 for l in out.splitlines() if l.startswith("{")) == 1    # ONE line, from rank 0
    # the N-rank line verifies itself: every rank contributed a digest (dry run: of the broadcast weight bytes), MIN == MAX
    assert line["rank_check"]["ranks_agree"] is True and "FAILED" not in line["rank_check"]
    assert len(line["init_s_per_rank"]) == 2 and all(v > 0 for v in line["init_s_per_rank"])


def test_a_rank_with_different_weights_fails_the_run():
    """--perturb-rank (TEST ONLY) makes rank 1 hold a slightly different weight tensor after the broadcast: the line must carry
    ranks_agree false and EVERY rank must exit non-zero -- an 8-GPU scaling run cannot pass with one rank computing something else."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--perturb-rank", "1"],
                       env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode != 0, out[-2000:]
    line = _last_json(out)
    assert line["rank_check"]["ranks_agree"] is False and line["rank_check"]["FAILED"] is True


def test_launcher_form_and_world_size_mismatch():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), BENCH]
    r = subprocess.run(base + ["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    assert _last_json(r.stdout.decode(errors="replace"))["world_size_seen"] == 2
    # the driver's launcher started 2 ranks but the flag says 4: refuse, do not print a line under the wrong label
    r = subprocess.run(base + ["--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode != 0 and not [l for l in out.splitlines() if l.startswith("{")], out[-2000:]


def test_single_rank_line_and_no_gpu_refusal():
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--dry-run"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    line = _last_json(r.stdout.decode(errors="replace"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "world_size_seen"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["world_size_seen"] == 1 and line["rccl_bcast_ms"] == 0.0
    # without --dry-run there is no CPU path: --gpus 2 on a box without 2 GPUs must exit non-zero instead of running one rank
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2"], env=_env(), cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode(errors="replace")
        assert r.returncode != 0 and "refusing" in out and not [l for l in out.splitlines() if l.startswith("{")]


def test_eight_ranks_the_size_of_the_scaling_run():
    """the first time this code meets 8 ranks must not be the driver's RCCL run.  The same dry run at world_size 8
    (the line carries every rank's CPU placement and host launch cost), and a perturbed LAST rank must fail all eight."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == 8 and line["world_size_seen"] == 8 and line["backend"] == "gloo"
    assert len(line["per_rank_fps"]) == 8 and len(line["init_s_per_rank"]) == 8
    assert len(line["host_launch_us_per_frame"]) == 8 and all(v >= 0 for v in line["host_launch_us_per_frame"])
    assert isinstance(line["cpu_affinity"], list) and len(line["cpu_affinity"]) == 8 and all("cpus" in s for s in line["cpu_affinity"])
    assert line["omp_num_threads"] >= 1
    assert line["rank_check"]["ranks_agree"] is True
    assert sum(1 for l in out.splitlines() if l.startswith("{")) == 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run", "--perturb-rank", "7"],
                       env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode != 0, out[-2000:]
    assert _last_json(out)["rank_check"]["ranks_agree"] is False


def test_eight_rank_dry_run_line_schema():
    """The WHOLE `bench.py --gpus 8` self-spawn path (launcher, rendezvous, pinning plan, weight broadcast, the solo reference run of rank 0, the
    8-rank timed loop, reductions, rank check) over gloo on the CPU, and the schema of the line the driver's scaling run will read: the
    collective block (backend, bytes, GB/s, hardware queues per rank), per-rank frames/s with min / max / spread, and the scaling check."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == 8 and line["world_size_seen"] == 8 and line["dry_run"] is True and line["value"] is None
    assert len(line["per_rank_fps"]) == 8 and len(line["init_s_per_rank"]) == 8 and len(line["host_launch_us_per_frame"]) == 8
    lo, hi, spread = line["per_rank_fps_min_max_spread"]
    assert lo == min(line["per_rank_fps"]) and hi == max(line["per_rank_fps"]) and 0.0 <= spread < 1.0
    c = line["collective"]
    assert c["backend"] == "gloo" and c["bcast_bytes"] == line["bcast_bytes"] > 1e6 and c["bcast_ms"] > 0 and c["bcast_GBps"] > 0
    assert len(c["hw_queues_per_rank"]) == 8 and all("GPU_MAX_HW_QUEUES" in q for q in c["hw_queues_per_rank"])
    sc = line["scaling_check"]
    assert set(sc) >= {"solo_fps_rank0", "per_gpu_fps", "scaling_efficiency", "what"} and sc["scaling_efficiency"] is None   # dry run: never a number
    assert line["rank_check"]["ranks_agree"] is True
    assert line["warmup"] >= line["warmup_requested"] == 1
    assert isinstance(line["cpu_affinity"], list) and len(line["cpu_affinity"]) == 8
