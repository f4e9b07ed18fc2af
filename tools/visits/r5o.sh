#!/bin/bash
# round 5, GPU visit o: the -m gpu suite alone on the final tree (visit e's last run: 55 passed + the idle-handle throughput probe's flake)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5o; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 2 $R/build.log
( time timeout 560 python -m pytest tests -q -m gpu -s --durations=8 ) > $R/gpu_tests.log 2>&1; tail -n 14 $R/gpu_tests.log
