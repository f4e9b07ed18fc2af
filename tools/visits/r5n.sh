#!/bin/bash
# round 5, GPU visit n: the queue check of the second chain's stream (td_frame.h place_chain_stream) under GPU_MAX_HW_QUEUES=4, verbose:
# which pairs of a busy handle's three streams share a hardware queue beside 0..3 idle handles, three runs
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r5n; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
for i in 1 2 3; do
  GPU_MAX_HW_QUEUES=4 TDNET_QUIET=1 TDNET_QUEUE_CHECK_VERBOSE=1 timeout 200 python tools/idle_handle_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $R/idle_probe_verbose.txt
done
