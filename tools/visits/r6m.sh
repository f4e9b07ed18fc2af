#!/bin/bash
# round 6, GPU visit m: precision 2 in the frame: persistent grid vs one tile per workgroup, chains vs none; fp16 gate with the per-class floor
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6m; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 3 $R/build.log
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 "precision=2" "precision=2,gemm_persistent=100000" "precision=2,overlap=40" "precision=2,overlap=40,gemm_persistent=100000" 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
timeout 900 python -m pytest tests -q -m gpu -s -k "test_fp16_model_gate or split_attention" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $R/tests.txt
