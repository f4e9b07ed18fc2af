#!/bin/bash
# Round-3 visit B: does a late-arriving transform kernel get admitted beside a persistent GEMM that leaves room (grid 512 = 2 per CU)?
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3b}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -s -k "uncalibrated or direct_conv" > $R/gpu_tests.log 2>&1
grep "reference init" $R/gpu_tests.log; tail -n 3 $R/gpu_tests.log
for v in "--overlap 0" "--overlap 0 --gemm-persistent 512" "--overlap 1 --gemm-persistent 512" "--overlap 33 --gemm-persistent 512" "--overlap 1 --gemm-persistent 640" "--overlap 33" "--overlap 0 --winograd 0"; do
  timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-pmc --no-direct-line $v > $R/v.log 2>&1
  echo "[$v] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt
done
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline --no-pmc --no-direct-line --overlap 1 --gemm-persistent 512"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$R/prof > $GRAFT_REPO_ROOT/$R/timeline.txt 2>&1
cd "$GRAFT_REPO_ROOT"
find $R -name "*.csv" -size +4M -delete
grep -n "q3" $R/timeline.txt | tail -12
