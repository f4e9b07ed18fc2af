#!/bin/bash
# Short GPU-box visit: kernel-trace profile of the default bench (per-kernel time only).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline $*"
rm -rf "$GRAFT_REPO_ROOT/$R/prof"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R 2>&1 | sed -n 1,40p
find $R -name "*.csv" -size +8M -delete
