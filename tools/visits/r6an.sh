#!/bin/bash
# round 6, GPU visit an: the persistent GEMM grid with precision 2 (gemm_persistent = workgroups per launch; 1 = the default 512 for k_gemm_b3) while two chains' GEMMs are in flight
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6an; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
python tools/ab_opts.py --size 1024x2048 --rounds 3 "precision=2" "precision=2,gemm_persistent=256" "precision=2,gemm_persistent=320" "precision=2,gemm_persistent=384" "precision=2,gemm_persistent=576" 2>&1 | tail -6 | tee $R/ab.txt
python tools/ab_opts.py --size 769x1537 --rounds 3 "precision=2" "precision=2,gemm_persistent=256" "precision=2,gemm_persistent=360" "precision=2,gemm_persistent=384" 2>&1 | tail -5 | tee -a $R/ab.txt
