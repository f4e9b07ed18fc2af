#!/bin/bash
# round 6, GPU visit g: grid size of the matrix-only split GEMM (persistent 512 workgroups vs one tile per workgroup vs in between)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6g; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for g in 1 512 576 640 768 1152 100000; do python tools/b3_one.py 5 512 512 4 20 $g; done 2>&1 | grep -v amdgpu.ids | tee -a $R/grid.txt
for g in 1 576 100000; do python tools/b3_one.py 5 256 256 2 20 $g; python tools/b3_one.py 5 256 512 4 20 $g; python tools/b3_one.py 5 512 128 1 20 $g; done 2>&1 | grep -v amdgpu.ids | tee -a $R/grid.txt
for g in 1 256 288 384 100000; do python tools/b3_one.py 3 512 512 4 20 $g; done 2>&1 | grep -v amdgpu.ids | tee -a $R/grid.txt
