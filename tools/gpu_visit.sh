#!/bin/bash
# One parameterised GPU-box visit (run through gpurun from the repo root); replaces the one-off gpu_r3*.sh scripts of round 3.
#   tools/gpu_visit.sh <tag> <step> [<step> ...]        outputs under gpurun_out/<tag>/; copy what is to be judged to profiles/
# steps (executed in order, each under its own timeout):
#   tests[:<pytest -k expression>]   the `-m gpu` suite (or the selected part)            -> gpu_tests.log
#   smoke                            __graft_entry__.smoke()                              -> smoke.log
#   bench[:<bench.py args>]          the default line with all legs + live PMC passes     -> bench_<n>.log / lines.jsonl
#   quick[:<bench.py args>]          bench.py --quick --steps 60 <args>                   -> lines.jsonl + summary.txt
#   ab:<ab_opts.py args>             same-process A/B of tdnet_opts variants (tools/ab_opts.py) -> ab.txt
#   prof:<name>[:<bench.py args>]    rocprofv3 --kernel-trace --stats of a short quick run -> kernel_stats_<name>.csv, timeline_<name>.txt
#   (counter passes: `cmd:tools/gpu_pmc.sh <tag2> <bench args>` -- one counter set per rocprofv3 pass, never with other trace domains)
#   cmd:<shell command>              anything else (probes)                               -> cmd_<n>.log
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-visit}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf "$R"; mkdir -p "$R"
n=0
line() { python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d.get("roofline",{})
    print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), "roofline", r.get("frac"), r.get("avg_launch_ms"), "sustained", d.get("sustained",{}).get("value"))
except Exception as e: print("FAILED", e)'; }
for step in "$@"; do
  n=$((n+1)); kind=${step%%:*}; rest=""; [ "$kind" != "$step" ] && rest=${step#*:}
  cd "$GRAFT_REPO_ROOT"
  case $kind in
    tests) if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -q -m gpu -s --durations=8 -k "$rest" > $R/gpu_tests_$n.log 2>&1; tail -n 6 $R/gpu_tests_$n.log
           else timeout 1700 python -m pytest tests -q -m gpu -s --durations=12 > $R/gpu_tests.log 2>&1; tail -n 16 $R/gpu_tests.log; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; tail -n 2 $R/smoke.log ;;
    bench) ( time timeout 900 python bench.py $rest ) > $R/bench_$n.log 2>&1; grep '^{' $R/bench_$n.log | tail -1 >> $R/lines.jsonl
           echo "[bench $rest] $(grep '^{' $R/bench_$n.log | tail -1 | line)" | tee -a $R/summary.txt ;;
    quick) timeout 300 python bench.py --steps 60 --quick $rest > $R/quick_$n.log 2>&1; grep '^{' $R/quick_$n.log | tail -1 >> $R/lines.jsonl
           echo "[quick $rest] $(grep '^{' $R/quick_$n.log | tail -1 | line)" | tee -a $R/summary.txt ;;
    ab)    eval "timeout 900 python tools/ab_opts.py --json $R/ab.jsonl $rest" 2>&1 | grep -v "amdgpu.ids" | tee -a $R/ab.txt ;;
    prof)  name=${rest%%:*}; args=""; [ "$name" != "$rest" ] && args=${rest#*:}
           ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/prof_$name" -o r1 -- \
               python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $args > "$R/prof_$name.log" 2>&1 )
           cp $(find $R/prof_$name -name "*kernel_stats.csv" | head -1) $R/kernel_stats_$name.csv 2>/dev/null
           python tools/timeline.py $R/prof_$name > $R/timeline_$name.txt 2>&1; head -n 12 $R/kernel_stats_$name.csv | cut -c1-150 ;;
    cmd)   eval "timeout 600 $rest" > $R/cmd_$n.log 2>&1; tail -n 40 $R/cmd_$n.log ;;
    *)     echo "unknown step $step" ;;
  esac
done
cd "$GRAFT_REPO_ROOT"
find $R -name "*kernel_trace.csv" -size +8M -delete; find $R -name "*.csv" -size +8M -delete; find $R -name "*.db" -delete
du -sh $R | tail -1
