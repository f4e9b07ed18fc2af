#!/bin/bash
# Round-3 visit Q: the next frame's cache-only chain launched at the end of the current frame (overlap bit 128).
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3q}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
run() { timeout 300 python bench.py --steps 60 $1 > $R/v.log 2>&1
  echo "[$1] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"), d.get("parity",{}).get("max_abs_dlogit"), d.get("parity",{}).get("flips_outside_tie_band"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt; tail -4 $R/v.log | head -3 | cut -c1-300 >> $R/errs.txt; }
run "--no-pmc --no-direct-line --no-other-configs --cpu-frames 2 --overlap 169"
run "--quick --overlap 41"; run "--quick --overlap 169"; run "--quick --overlap 41"; run "--quick --overlap 169"
run "--quick --model td2 --size 1024x2048 --overlap 41"; run "--quick --model td2 --size 1024x2048 --overlap 169"
run "--quick --model td4 --size 1024x2048 --precision fp16 --overlap 128"; run "--quick --model td4 --size 1024x2048 --precision fp16"
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --overlap 169"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$R/prof > $GRAFT_REPO_ROOT/$R/timeline.txt 2>&1
cd "$GRAFT_REPO_ROOT"; find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
grep "k_conv_igemm<128, 64\|k_attention<1, 4, 2" $R/timeline.txt | head -12
