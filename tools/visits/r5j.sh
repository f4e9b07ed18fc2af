#!/bin/bash
# round 5, GPU visit j: the 7x7 stem on the packed-row image (fusion bit 65536): A/B by size, parity block, kernel time
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5j; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu 2>&1 | tail -2
for sz in 512x1024 769x1537 1024x2048; do
  timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --size $sz --steps 80 --rounds 3 "" "fusion=106534" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp32_stem_rows_by_size.txt
done
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --size 1024x2048 --steps 80 --rounds 3 "" "fusion=106534" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp32_stem_rows_by_size.txt
timeout 600 python bench.py --fusion 106534 --no-pmc --no-direct-line --no-other-configs --steps 40 2>/dev/null | grep '^{' | tail -1 > $R/line_stem_rows.json
python -c "
import json; d=json.load(open('$R/line_stem_rows.json')); print(d['value'], d['parity'])"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --fusion 106534 > $R/prof.log 2>&1 )
grep -h "adirect<7\|rgbpad\|nhwc4" $(find $R/prof -name "*kernel_stats.csv" | head -1) | cut -c1-150
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
