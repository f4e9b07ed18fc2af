#!/bin/bash
# round 5, GPU visit d: kernel stats of the Bottleneck models at the reference's native 769x1537 (N1 td2-psp50, N3 psp101) + the changed GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5d; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
for cfg in "psp101:--model psp --backbone resnet101 --size 769x1537" "td2psp50:--model td2 --backbone resnet50 --size 769x1537" "td4_769:--size 769x1537"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$name -o r1 -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $args > $R/prof_$name.log 2>&1 )
  cp $(find $R/prof_$name -name "*kernel_stats.csv" | head -1) $R/kernel_stats_$name.csv 2>/dev/null
  echo "== $name"; grep '^{' $R/prof_$name.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'],'fps (profiled)', d['frame']['executed_gflop'],'executed GFLOP', d['frame']['algorithmic_gflop'], 'algorithmic', d['breakdown_ms_per_frame'], d['launches_per_frame'])"
  head -n 16 $R/kernel_stats_$name.csv | cut -c1-150
done
echo "== changed GPU tests"
timeout 900 python -m pytest tests -q -m gpu -k "fp16_kernels or full_size_digest or batch_of_streams or every_schedule" 2>&1 | tail -4
find $R -name "*kernel_trace.csv" -size +6M -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
