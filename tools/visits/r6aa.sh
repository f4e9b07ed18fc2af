#!/bin/bash
# round 6, GPU visit aa: the precision-2 gates with layer1 + stem on the split kernel (tests/test_gpu_b3.py, all of it), kernel stats + timeline of the precision-2 frame
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
bash tools/gpu_visit.sh r6aa "tests:test_gpu_b3" "prof:b3:--precision bf16x3"
