#!/bin/bash
# GPU-box visit: Winograd bring-up.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out
timeout 300 python tools/wino_probe.py > $R/wino_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_winograd.py -q -m gpu -s 2>&1 | tail -25 > $R/wino_test.log
timeout 300 python bench.py --steps 40 --winograd 1 > $R/bench_wino1.log 2>&1
timeout 300 python bench.py --steps 40 --winograd 2 --no-cpu-baseline > $R/bench_wino2.log 2>&1
timeout 300 python bench.py --steps 40 --size 769x1537 --winograd 1 > $R/bench_native_wino1.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/profw" -o r1 -- python $GRAFT_REPO_ROOT/bench.py --winograd 1 --steps 8 --warmup 6 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$R/profw.log" 2>&1
cd "$GRAFT_REPO_ROOT"
sed -i 's|for sub in ("prof", "prof50",|for sub in ("prof", "profw", "prof50",|' tools/summarize_prof.py
python tools/summarize_prof.py $R > $R/prof_summary.txt 2>&1
find $R -name "*.csv" -size +8M -delete
tail -12 $R/wino_probe.log; tail -8 $R/wino_test.log
