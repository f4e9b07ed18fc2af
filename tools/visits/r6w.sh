#!/bin/bash
# round 6, GPU visit w: counters of k_conv_adirect_b3 on layer1's conv
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6w; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
PMC_CMD="python $GRAFT_REPO_ROOT/tools/adb3_probe.py 256x512 10 split" bash tools/b3_pmc.sh $R/pmc_split 2>&1 | tee $R/pmc_split.txt
PMC_CMD="python $GRAFT_REPO_ROOT/tools/adb3_probe.py 256x512 10 fp32" bash tools/b3_pmc.sh $R/pmc_fp32 2>&1 | tee $R/pmc_fp32.txt
