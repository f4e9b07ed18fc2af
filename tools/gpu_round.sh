#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Everything is logged under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/ops.log
echo "ops rc=$?" >> gpurun_out/ops.log
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu -s 2>&1 | tail -60 > gpurun_out/model.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
timeout 600 python bench.py --model td2 --steps 40 > gpurun_out/bench_td2.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 6 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof -name "*kernel_stats*" | head; find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
tail -5 gpurun_out/ops.log gpurun_out/model.log gpurun_out/smoke.log gpurun_out/bench.log
