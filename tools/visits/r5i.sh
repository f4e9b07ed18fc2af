#!/bin/bash
# round 5, GPU visit i: Winograd F(4x4) for ResNet layer1 too (winograd = 4) by map size, with the parity block of bench.py at 1024x2048
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5i; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
for sz in 512x1024 769x1537 1024x2048; do
  timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --size $sz --steps 80 --rounds 3 "" "winograd=4" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp32_winograd4_by_size.txt
done
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --size 1024x2048 --steps 80 --rounds 3 "" "winograd=4" 2>&1 | grep -v amdgpu.ids | tee -a $R/ab_fp32_winograd4_by_size.txt
timeout 600 python bench.py --winograd 4 --no-pmc --no-direct-line --no-other-configs --steps 40 2>/dev/null | grep '^{' | tail -1 > $R/line_winograd4.json
python -c "
import json; d=json.load(open('$R/line_winograd4.json')); print(d['value'], d['parity'])"
du -sh $R | tail -1
