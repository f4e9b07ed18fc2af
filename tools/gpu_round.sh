#!/bin/bash
# GPU-box visit (run through gpurun from the repo root): parity suite, smoke, the default bench line (incl. its live PMC passes),
# and a rocprofv3 kernel trace of the same command.  Everything lands under gpurun_out/<tag>/; copy what is to be judged to profiles/.
#   tools/gpu_round.sh <tag> [tests|notests] [extra bench args...]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-round}; shift
TESTS=${1:-tests}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
if [ "$TESTS" = "tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -s --durations=12 > $R/gpu_tests.log 2>&1
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1
fi
timeout 900 python bench.py "$@" > $R/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline --no-pmc --no-direct-line $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R > $R/prof_summary.txt 2>&1
cp $(find $R/prof -name "*kernel_stats.csv" | head -1) $R/kernel_stats.csv 2>/dev/null
find $R -name "*.csv" -size +4M -delete
tail -n 8 $R/gpu_tests.log 2>/dev/null; tail -n 3 $R/smoke.log 2>/dev/null; tail -n 1 $R/bench.log
