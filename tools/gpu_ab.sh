#!/bin/bash
# GPU-box visit: A/B of the per-handle kernel options on the headline workload (short timed regions, no CPU baseline / PMC), then probes.
#   tools/gpu_ab.sh <tag> "<bench args of variant 1>" "<bench args of variant 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ab}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
i=0
for v in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-pmc --no-direct-line $v > $R/v$i.log 2>&1
  echo "[$v] $(tail -1 $R/v$i.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt
done
timeout 200 python tools/attn_probe.py > $R/attn_probe.log 2>&1; cat $R/attn_probe.log
