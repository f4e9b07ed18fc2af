#!/bin/bash
# round 6, GPU visit t: k_gemm_b3 timeline with the A / B traffic switched off in turn
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6t; rm -rf "$R"; mkdir -p "$R"
for d in 0 1 2 3; do tools/_build/gemm_b3_trace 2048 512 512 0 $d > $R/trace_dead$d.txt; head -9 $R/trace_dead$d.txt; done
