#!/bin/bash
# round 6, GPU visit j: GPU tests of precision 2 (operators, fp32 gates on calibrated and un-calibrated weights)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
bash tools/gpu_visit.sh r6j "tests:test_gpu_b3"
grep -i "x[0-9]\.\|pooled\|conv \|worst" gpurun_out/r6j/gpu_tests_1.log | cut -c1-260
