#!/bin/bash
# round 5, GPU visit m: the pyramid conv with its weights requested ahead of the pooling phase, now with launch bounds (visit e's build had
# spilled: 128 VGPRs under the default 1024-thread bound, 13.5 -> 45 us): kernel durations in the fp16 720x960 and the fp32 default frame
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5m; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
for cfg in "fp16_720:--model td2 --backbone resnet34 --size 720x960 --precision fp16" "fp32_default:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$name -o r1 -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $args > $R/prof_$name.log 2>&1 )
  cp $(find $R/prof_$name -name "*kernel_stats.csv" | head -1) $R/kernel_stats_$name.csv 2>/dev/null
  echo "== $name"; grep -h "k_ppm_pool_conv\|k_ppm_rowbins\|k_ln_finalize" $R/kernel_stats_$name.csv | cut -c1-130
done
timeout 300 python bench.py --steps 60 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16 720x960', d['value'], d['launches_per_frame'])"
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.db" -delete
