#!/bin/bash
# round 6, GPU visit a: first run of the bf16x3 split GEMM (precision 2): accuracy + per-conv timing, kernel stats, frame A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6a; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 2 $R/build.log
timeout 600 python tools/b3_probe.py > $R/b3_probe.txt 2>&1; cat $R/b3_probe.txt | grep -v amdgpu.ids
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/prof_b3" -o r1 -- python $GRAFT_REPO_ROOT/tools/b3_probe.py > "$R/prof_b3.log" 2>&1 )
cp $(find $R/prof_b3 -name "*kernel_stats.csv" | head -1) $R/kernel_stats_b3_probe.csv 2>/dev/null; head -n 14 $R/kernel_stats_b3_probe.csv | cut -c1-160
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 "" "precision=2" "precision=2,overlap=40" "overlap=40" 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
rm -rf $R/prof_b3
