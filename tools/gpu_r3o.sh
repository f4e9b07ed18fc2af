#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out/${1:-r3o}; rm -rf $R; mkdir -p $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2; do
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --overlap 105"
TD_RIDER_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof$dbg" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof$dbg.log" 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/$R/prof$dbg > $GRAFT_REPO_ROOT/$R/timeline$dbg.txt 2>&1
echo "== TD_RIDER_DEBUG=$dbg"; grep "k_gemm_dma" $GRAFT_REPO_ROOT/$R/timeline$dbg.txt | sed -n 22,30p
done
cd "$GRAFT_REPO_ROOT"; find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
