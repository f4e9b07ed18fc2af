#!/bin/bash
# round 5, GPU visit c: fp16 tile-routing A/B (narrow tiles for 512 channels, 64- and 96-row narrow tiles)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5c; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 2 $R/build.log
echo "== A/B fp16 td2-psp34 720x960: tile routing"
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --backbone resnet34 --size 720x960 --precision fp16 --steps 80 --rounds 3 \
    "" "fusion=106534" "fusion=630822" "fusion=172070" "fusion=303142" "fusion=499750" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp16_720_tiles.txt
echo "== isolated conv timings (tdnet_op_conv2d_f16io through opcheck is for parity; timing: rocprof of one A/B round)"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_all -o r1 -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion 499750 > $R/prof_all.log 2>&1 )
cp $(find $R/prof_all -name "*kernel_stats.csv" | head -1) $R/kernel_stats_fp16_720_all_hooks.csv 2>/dev/null; head -n 14 $R/kernel_stats_fp16_720_all_hooks.csv | cut -c1-140
python tools/timeline.py $R/prof_all > $R/timeline_fp16_720_all_hooks.txt 2>&1; head -n 1 $R/timeline_fp16_720_all_hooks.txt
echo "== A/B fp16 td4-psp18 1024x2048 (the hooks only apply to maps <= 16384 px: must be unchanged)"
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --precision fp16 --steps 40 --rounds 2 "" "fusion=499750" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp16_c3.txt
find $R -name "*kernel_trace.csv" -size +6M -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
