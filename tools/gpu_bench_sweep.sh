#!/bin/bash
# GPU-box visit: bench lines of every workload (with CPU baseline + parity where stated) for the record.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=gpurun_out/sweep
rm -rf $R; mkdir -p $R
timeout 600 python bench.py > $R/01_td4_c3.log 2>&1
timeout 300 python bench.py --model td2 --steps 40 > $R/02_td2_c2.log 2>&1
timeout 300 python bench.py --size 769x1537 --steps 40 > $R/03_td4_native.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet50 --size 769x1537 --steps 40 > $R/04_td2psp50.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 > $R/05_td2psp34.log 2>&1
timeout 300 python bench.py --model psp --size 769x1537 --steps 30 --cpu-frames 1 > $R/06_psp101.log 2>&1
timeout 300 python bench.py --winograd 0 --steps 40 --no-cpu-baseline > $R/07_td4_direct.log 2>&1
timeout 300 python bench.py --winograd 1 --steps 40 --no-cpu-baseline > $R/08_td4_f2.log 2>&1
timeout 300 python bench.py --clips-per-gpu 3 --steps 30 --no-cpu-baseline > $R/09_td4_3clips.log 2>&1
timeout 300 python bench.py --mode path-parallel --steps 40 --no-cpu-baseline > $R/10_td4_pathparallel_n1.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 --precision fp16 > $R/11_td2psp34_fp16.log 2>&1
timeout 300 python bench.py --precision fp16 --steps 40 > $R/12_td4_fp16.log 2>&1
for f in $R/*.log; do tail -1 $f; done > $R/lines.jsonl
wc -l $R/lines.jsonl
