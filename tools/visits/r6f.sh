#!/bin/bash
# round 6, GPU visit f: counters of the split GEMM kernels on layer4's 512 -> 512 conv
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6f; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for t in 5 3 -2; do bash tools/b3_pmc.sh $R/t$t $t 512 512 4 10 2>&1 | grep -v amdgpu.ids | tee -a $R/pmc.txt; done
rm -rf $R/t*/p*/
