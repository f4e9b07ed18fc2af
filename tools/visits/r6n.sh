#!/bin/bash
# round 6, GPU visit n: same-process A/B of the two forms of the bf16x3 attention kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6n; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 3 $R/build.log
python tools/attn_b3_ab.py 1024x2048 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
python tools/attn_b3_ab.py 769x1537 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
