#!/bin/bash
# Round-3 final visit: GPU parity suite, smoke, the default bench line (all legs, live PMC), quick lines of the other workloads,
# rocprofv3 kernel stats + one-frame timeline of the default, kernel stats of the two fp16 workloads.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3final}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $R/gpu_tests.log 2>&1; tail -n 10 $R/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; tail -n 2 $R/smoke.log
( time timeout 600 python bench.py ) > $R/bench_default.log 2>&1; tail -n 4 $R/bench_default.log | cut -c1-250
run() { timeout 300 python bench.py --steps 60 --quick $1 > $R/v.log 2>&1
  echo "[$1] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt; tail -1 $R/v.log >> $R/lines.jsonl; }
run ""; run "--overlap 0"; run ""
run "--model td2 --size 1024x2048"; run "--model td4 --size 769x1537"; run "--model td2 --backbone resnet50 --size 769x1537"
run "--model psp --size 769x1537"; run "--model td4 --backbone resnet34 --size 1024x2048"; run "--model td4 --backbone resnet50 --size 769x1537"
run "--model td2 --backbone resnet34 --size 720x960"; run "--model td2 --backbone resnet34 --size 720x960 --precision fp16"; run "--model td4 --size 1024x2048 --precision fp16"
run "--model td4 --size 769x1537 --precision fp16"; run "--model td2 --size 1024x2048 --precision fp16"; run "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 2086"; run "--model td4 --size 1024x2048 --precision fp16 --fusion 2086"; run "--fusion 6"; run ""
cd /tmp && export TMPDIR=/tmp
prof() { timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/$1" -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $2 > "$GRAFT_REPO_ROOT/$R/$1.log" 2>&1
  cp $(find $GRAFT_REPO_ROOT/$R/$1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$R/kernel_stats_$1.csv 2>/dev/null; }
prof prof ""
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$R/prof > $GRAFT_REPO_ROOT/$R/timeline.txt 2>&1
prof prof_fp16_td4 "--model td4 --size 1024x2048 --precision fp16"
prof prof_fp16_td2_720 "--model td2 --backbone resnet34 --size 720x960 --precision fp16"
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R > $R/prof_summary.txt 2>&1
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
