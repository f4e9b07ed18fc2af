#!/bin/bash
# round 6, GPU visit r: k_gemm_b3 with three A buffers (A(g + 3) in flight across the barrier) against two
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6r; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 3 $R/build.log
for i in 1 2; do python tools/b3_one.py 5 512 512 4 20; TD_B3_NA3=1 python tools/b3_one.py 5 512 512 4 20; done 2>&1 | grep -v amdgpu.ids | tee $R/one.txt
python tools/b3_one.py 5 256 256 2 20 2>&1 | grep -v amdgpu.ids | tee -a $R/one.txt; TD_B3_NA3=1 python tools/b3_one.py 5 256 256 2 20 2>&1 | grep -v amdgpu.ids | tee -a $R/one.txt
python tools/env_ab.py TD_B3_NA3 1024x2048 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
python tools/env_ab.py TD_B3_NA3 769x1537 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
