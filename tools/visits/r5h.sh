#!/bin/bash
# round 5, GPU visit h: the option set at the reference's native 769x1537 (every round tuned at 1024x2048 only)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5h; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --size 769x1537 --steps 80 --rounds 3 "" "overlap=45" "overlap=42" "overlap=8" "winograd=4" "attention=1" "attention=0" \
    "fusion=41014" "fusion=41062" "fusion=40994" "fusion=40966" "gemm_persistent=0" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp32_769_options.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_769 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --size 769x1537 > $R/prof_769.log 2>&1 )
cp $(find $R/prof_769 -name "*kernel_stats.csv" | head -1) $R/kernel_stats_769.csv; head -n 24 $R/kernel_stats_769.csv | cut -c1-130
python tools/timeline.py $R/prof_769 > $R/timeline_769.txt 2>&1; head -1 $R/timeline_769.txt
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
