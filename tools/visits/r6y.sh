#!/bin/bash
# round 6, GPU visit y: leave-one-out timing of k_conv_adirect_b3 (tools/adb3_skip_probe.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r6y; rm -rf "$R"; mkdir -p "$R"
for rep in 1 2; do for m in 0 1 2 4 8 16 31; do tools/_build/adb3_skip_$m; done; for o in 2 3; do echo -n "occupancy $o: "; tools/_build/adb3_occ_$o; done; done 2>&1 | tee $R/skip.txt
