#!/bin/bash
# Round-3 visit R: ring depth of the 128 x 128 LDS-DMA conv (fp16 mode) where the grid does not fill the chip; fp16 parity tests; fp16 lines.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3r}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 600 python -m pytest tests/test_gpu_fp16.py -x -q > $R/fp16_tests.log 2>&1; tail -3 $R/fp16_tests.log
bash tools/conv_h_ring_probe.sh $TAG/ring > /dev/null 2>&1; cat $R/ring/summary.txt
run() { timeout 300 python bench.py --steps 60 $1 > $R/v.log 2>&1
  echo "[$1] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"), d.get("parity",{}).get("max_abs_dlogit"), d.get("parity",{}).get("flips_outside_tie_band"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt; tail -4 $R/v.log | head -3 | cut -c1-300 >> $R/errs.txt; }
run "--quick --model td2 --backbone resnet34 --size 720x960 --precision fp16"; run "--quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 2054"
run "--quick --model td4 --size 1024x2048 --precision fp16"
run "--quick --model td4 --size 769x1537 --precision fp16"; run "--no-pmc --no-direct-line --no-other-configs --cpu-frames 2 --model td4 --size 1024x2048 --precision fp16"
run "--quick --model td2 --backbone resnet34 --size 720x960 --precision fp16"; run "--quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 2054"
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$R/prof > $GRAFT_REPO_ROOT/$R/timeline.txt 2>&1
cd "$GRAFT_REPO_ROOT"; cp $(find $R/prof -name "*kernel_stats.csv" | head -1) $R/kernel_stats_720.csv 2>/dev/null
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
