#!/bin/bash
# Round-3 visit A: GPU parity suite on the new default (row-parity chains), then an A/B of tdnet_opts.overlap variants on the headline
# workload, then a kernel trace of the default for the timeline.   tools/gpu_r3a.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3a}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 900 python -m pytest tests -q -m gpu -x --durations=8 > $R/gpu_tests.log 2>&1
tail -n 6 $R/gpu_tests.log
for v in "--overlap 0" "--overlap 2" "--overlap 18" "--overlap 34" "--overlap 1" "--overlap 17" "--overlap 33" "--overlap 0" "--overlap 1"; do
  timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-pmc --no-direct-line $v > $R/v.log 2>&1
  echo "[$v] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt
  tail -3 $R/v.log | head -2 | cut -c1-300 >> $R/errs.txt
done
cd /tmp && export TMPDIR=/tmp
for ov in 1 0; do
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline --no-pmc --no-direct-line --overlap $ov"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof$ov" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof$ov.log" 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$R/prof$ov > $GRAFT_REPO_ROOT/$R/timeline$ov.txt 2>&1
cp $(find $GRAFT_REPO_ROOT/$R/prof$ov -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$R/kernel_stats$ov.csv 2>/dev/null
done
cd "$GRAFT_REPO_ROOT"
find $R -name "*.csv" -size +4M -delete
tail -n 4 $R/timeline1.txt
