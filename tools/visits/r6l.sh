#!/bin/bash
# round 6, GPU visit l: the bf16x3 attention kernel: accuracy, kernel durations, frame A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6l; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 3 $R/build.log
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/prof" -o r1 -- python $GRAFT_REPO_ROOT/tools/attn_b3_probe.py > "$R/attn_b3_probe.txt" 2>&1 )
grep -v "amdgpu.ids\|^W2026\|^E2026" $R/attn_b3_probe.txt
cp $(find $R/prof -name "*kernel_stats.csv" | head -1) $R/kernel_stats_attn.csv 2>/dev/null; grep "k_att" $R/kernel_stats_attn.csv | cut -c1-160
rm -rf $R/prof
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 3 "" "precision=2" 2>&1 | grep -v amdgpu.ids | tee $R/ab.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 2 --size 769x1537 "" "precision=2" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --rounds 2 --model td2 "" "precision=2" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab.txt
