#!/bin/bash
# round 6, GPU visit h: the frame with the matrix-only split GEMM (precision 2), with and without the row-parity chains
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6h; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
python tools/b3_one.py 5 512 512 4 20 2>&1 | grep -v amdgpu.ids | tee $R/one.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 "" "precision=2" "precision=2,overlap=40" 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 2 --size 769x1537 "" "precision=2" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 2 --model td2 "" "precision=2" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
