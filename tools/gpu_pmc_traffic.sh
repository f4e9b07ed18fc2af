#!/bin/bash
# GPU-box visit: FETCH_SIZE / WRITE_SIZE passes of the default bench (HBM traffic of the dominant kernel), one counter per pass.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline"
rm -rf "$GRAFT_REPO_ROOT/$R/pmc_fetch" "$GRAFT_REPO_ROOT/$R/pmc_write" "$GRAFT_REPO_ROOT/$R/pmc_sq" "$GRAFT_REPO_ROOT/$R/prof"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/$R/pmc_fetch" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/$R/pmc_write" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R 2>&1 | grep "k_gemm_persistent\|k_ln_partial\|k_wino4\|====" 
find $R -name "*.csv" -size +8M -delete
