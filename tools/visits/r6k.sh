#!/bin/bash
# round 6, GPU visit k: the whole -m gpu suite + the full default bench line with the new legs (precision 2, hipgraph, numerics_stress)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
bash tools/gpu_visit.sh r6k "tests" "smoke" "bench:--steps 20 --warmup 5" "bench:--steps 40 --warmup 8 --graph --quick" "bench:--steps 40 --warmup 8 --precision bf16x3 --no-other-configs --no-numerics-stress"
