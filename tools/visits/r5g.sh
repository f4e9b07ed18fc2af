#!/bin/bash
# round 5, GPU visit g: at which map size do the fp32 row-parity chains start to pay?  (default overlap 41 vs 40 = the same kernels, no chains)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5g; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
for sz in 512x1024 640x1280 769x1537 896x1792 1024x2048; do
  timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --size $sz --steps 80 --rounds 3 "" "overlap=40" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp32_chains_by_size.txt
done
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --size 769x1537 --steps 80 --rounds 3 "" "overlap=40" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp32_chains_by_size.txt
du -sh $R | tail -1
