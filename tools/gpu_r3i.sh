#!/bin/bash
# Round-3 visit I: fp16 lines after the DMA issue was interleaved with the MFMAs; per-kernel stats at both sizes; fp16 parity tests.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3i}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 300 python -m pytest tests/test_gpu_fp16.py -q -m gpu -x -s > $R/gpu_fp16.log 2>&1; tail -n 4 $R/gpu_fp16.log
run() { timeout 200 python bench.py --steps 60 --quick $1 > $R/v.log 2>&1
  echo "[$1] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt; tail -4 $R/v.log | head -3 | cut -c1-300 >> $R/errs.txt; }
run "--model td4 --size 1024x2048 --precision fp16"
run "--model td4 --size 1024x2048 --precision fp16 --fusion 134"
run "--model td2 --backbone resnet34 --size 720x960 --precision fp16"
run "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 134"
run "--model td2 --backbone resnet18 --size 1024x2048 --precision fp16"
cd /tmp && export TMPDIR=/tmp
for cfg in "td4 resnet18 1024x2048 6" "td2 resnet34 720x960 6" "td2 resnet34 720x960 134"; do set -- $cfg
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model $1 --backbone $2 --size $3 --precision fp16 --fusion $4"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof_$1_$4" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof_$1_$4.log" 2>&1
cp $(find $GRAFT_REPO_ROOT/$R/prof_$1_$4 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$R/kernel_stats_fp16_$1_$3_fusion$4.csv 2>/dev/null
done
cd "$GRAFT_REPO_ROOT"
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
