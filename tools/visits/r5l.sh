#!/bin/bash
# round 5, GPU visit l: grouped launches of the register-staged fp16 convs (fusion bit 131072), deeper unrolling in the pyramid conv /
# LayerNorm finalize / classifier: A/B in the fp16 mode, per-kernel durations, fp32 headline check
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5l; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp16.py -q -m gpu -x 2>&1 | tail -3
G=$((106534 | 131072))
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --backbone resnet34 --size 720x960 --precision fp16 --steps 200 --rounds 4 "" "fusion=$G" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp16_grouped_launches.txt
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --size 1024x2048 --precision fp16 --steps 100 --rounds 3 "" "fusion=$G" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp16_grouped_launches.txt
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --size 1024x2048 --steps 80 --rounds 2 "" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp16_grouped_launches.txt
timeout 600 python bench.py --model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion $G --no-pmc --no-direct-line --no-other-configs --no-cpu-baseline --steps 60 2>/dev/null | grep '^{' | tail -1 > $R/line_fp16_grouped.json
python -c "
import json; d=json.load(open('$R/line_fp16_grouped.json')); print(d['value'], d['launches_per_frame'], d['roofline']['frac'], d['parity'])"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion $G --steps 8 --warmup 6 --quick > $R/prof.log 2>&1 )
T=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $R/prof 2 > $R/timeline_fp16_td2psp34_720x960_grouped.txt 2>&1; head -70 $R/timeline_fp16_td2psp34_720x960_grouped.txt
cp $(find $R/prof -name "*kernel_stats.csv" | head -1) $R/kernel_stats_fp16_720x960_grouped.csv
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
