#!/bin/bash
# round 6, GPU visit ac: precision-2 gates with the 64-query attention form picked by size; frame numbers next to the default, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6ac; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_b3.py -x -q -s > $R/test_b3.log 2>&1; grep -E "attention L|pooled|769x1537 bf16x3|passed|failed|Error|assert" $R/test_b3.log | tail -16
python tools/ab_opts.py --size 1024x2048 --rounds 3 "" "precision=2" 2>&1 | tail -3 | tee $R/ab_1024.txt
python tools/ab_opts.py --size 769x1537 --rounds 3 "" "precision=2" 2>&1 | tail -3 | tee $R/ab_769.txt
python tools/ab_opts.py --model td2 --backbone resnet18 --size 1024x2048 --rounds 2 "" "precision=2" 2>&1 | tail -3 | tee $R/ab_td2.txt
