#!/bin/bash
# Round-3 visit E: whole GPU suite on HEAD (chains with 4 channels per lane as the default again, TD_VW4_FIX = 1), PMC on the fp16 conv
# kernels, fp16 lines with the new tile heuristic, the default bench line (all legs) with its wall time.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3e}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $R/gpu_tests.log 2>&1; tail -n 14 $R/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; tail -n 2 $R/smoke.log
run() { timeout 200 python bench.py --steps 60 --quick $1 > $R/v.log 2>&1
  echo "[$1] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt; tail -4 $R/v.log | head -3 | cut -c1-300 >> $R/errs.txt; }
run "--model td4 --size 1024x2048 --precision fp16"
run "--model td2 --backbone resnet34 --size 720x960 --precision fp16"
run "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 134"
run ""; run "--overlap 1"; run "--overlap 0"
( time timeout 600 python bench.py ) > $R/bench_default.log 2>&1; tail -n 5 $R/bench_default.log | cut -c1-200
bash tools/conv_h_pmc.sh $TAG/convh_pmc
