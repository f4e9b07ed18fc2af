#!/bin/bash
# round 6, GPU visit p: the classifier inside the head's output transform (fusion bit 262144): frame A/B, fp32 and precision 2, two sizes
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6p; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 3 $R/build.log
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 4 "" "fusion=499750" "precision=2" "precision=2,fusion=499750" 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 --size 769x1537 "" "fusion=499750" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
