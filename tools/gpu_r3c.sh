#!/bin/bash
# Round-3 visit C: LDS-DMA probe, whole GPU parity suite on HEAD, fp16 A/B (LDS-DMA conv kernel vs the register-staged one: fusion 128),
# the default bench line with its other_configs legs, kernel trace of the fp16 workloads.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3c}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 60 tools/_build/lds_dma_probe 2>&1 | tee $R/lds_dma_probe.txt
timeout 300 python -m pytest tests/test_gpu_fp16.py -q -m gpu -x -s > $R/gpu_fp16.log 2>&1; tail -n 5 $R/gpu_fp16.log
for v in "--model td4 --size 1024x2048 --precision fp16" "--model td4 --size 1024x2048 --precision fp16 --fusion 134" \
         "--model td2 --backbone resnet34 --size 720x960 --precision fp16" "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 134"; do
  timeout 200 python bench.py --steps 60 --quick $v > $R/v.log 2>&1
  echo "[$v] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt
  tail -4 $R/v.log | head -3 | cut -c1-400 >> $R/errs.txt
done
timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $R/gpu_tests.log 2>&1; tail -n 12 $R/gpu_tests.log
( time timeout 600 python bench.py ) > $R/bench_default.log 2>&1; tail -n 5 $R/bench_default.log | cut -c1-3000
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof_fp16" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
cp $(find $GRAFT_REPO_ROOT/$R/prof_fp16 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$R/kernel_stats_fp16_720x960.csv 2>/dev/null
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model td4 --size 1024x2048 --precision fp16"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof_fp16b" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/profb.log" 2>&1
cp $(find $GRAFT_REPO_ROOT/$R/prof_fp16b -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$R/kernel_stats_fp16_1024x2048.csv 2>/dev/null
cd "$GRAFT_REPO_ROOT"
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
head -12 $R/kernel_stats_fp16_1024x2048.csv | cut -c1-200
