#!/bin/bash
# One GPU-box visit (this variant: fp16-MFMA mode bring-up + probes).  Everything is logged under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out
timeout 600 python -m pytest tests/test_gpu_fp16.py -q -m gpu -s 2>&1 | tail -25 > $R/fp16.log
timeout 400 python tools/kernel_probe.py > $R/probe.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 --precision fp16 > $R/bench_td2psp34_fp16.log 2>&1
timeout 300 python bench.py --steps 40 --precision fp16 > $R/bench_td4_fp16.log 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -5 > $R/ops.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof16" -o r1 -- python $GRAFT_REPO_ROOT/bench.py --model td2 --backbone resnet34 --size 720x960 --precision fp16 --steps 8 --warmup 6 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$R/prof16.log" 2>&1
cd "$GRAFT_REPO_ROOT"
sed -i 's|for sub in ("prof", "prof50",|for sub in ("prof", "prof50", "prof16",|' tools/summarize_prof.py
python tools/summarize_prof.py $R > $R/prof_summary.txt 2>&1
find $R -name "*.csv" -size +8M -delete
tail -5 $R/fp16.log $R/ops.log
