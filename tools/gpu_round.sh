#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench lines (td4 C3, td2, native size, td2-psp50), rocprof kernel trace.
# PMC passes: tools/gpu_pmc.sh.  Everything is logged under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -15 > $R/ops.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_psp101.py tests/test_gpu_harness.py -q -m gpu -s 2>&1 | tail -30 > $R/model.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1
timeout 600 python bench.py > $R/bench.log 2>&1
timeout 300 python bench.py --model td2 --steps 40 > $R/bench_td2.log 2>&1
timeout 300 python bench.py --size 769x1537 --steps 40 > $R/bench_native.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet50 --size 769x1537 --steps 40 > $R/bench_td2psp50.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 > $R/bench_td2psp34.log 2>&1
timeout 300 python bench.py --model psp --size 769x1537 --steps 30 --cpu-frames 1 > $R/bench_psp101.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof50" -o r1 -- python $GRAFT_REPO_ROOT/bench.py --model td2 --backbone resnet50 --size 769x1537 --steps 8 --warmup 6 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$R/prof50.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R > $R/prof_summary.txt 2>&1
find $R -name "*.csv" -size +8M -delete
tail -3 $R/ops.log $R/model.log $R/smoke.log $R/bench.log
