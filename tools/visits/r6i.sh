#!/bin/bash
# round 6, GPU visit i: kernel stats + timeline of the precision-2 frame (what is left beside the split GEMMs), and the hipGraph tests
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
bash tools/gpu_visit.sh r6i "prof:b3:--precision bf16x3" "prof:fp32" "tests:test_gpu_graph"
