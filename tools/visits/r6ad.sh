#!/bin/bash
# round 6, GPU visit ad: k_conv_adirect_b3 with the step's vector-memory requests spread over the MFMA groups (TD_ADB3_SPREAD=1) against all in group 0
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT/gpurun_out/r6ad; rm -rf "$R"; mkdir -p "$R"
for rep in 1 2 3; do for sp in 0 1; do echo -n "spread $sp: "; tools/_build/adb3_spread_$sp; done; done 2>&1 | tee $R/spread.txt
for sp in 0 1; do echo -n "193x385 spread $sp: "; tools/_build/adb3_spread_$sp 193 385; done 2>&1 | tee -a $R/spread.txt
