#!/bin/bash
# round 6, GPU visit q: Encoding's q / k branches on the side stream (fusion bit 1) re-measured on the round-6 frame, fp32 and precision 2
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6q; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 3 $R/build.log
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 "" "fusion=499751" "precision=2" "precision=2,fusion=499751" 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 --size 769x1537 "" "fusion=499751" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
