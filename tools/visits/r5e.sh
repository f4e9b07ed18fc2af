#!/bin/bash
# round 5, GPU visit e (final set): full -m gpu suite, smoke, the full default bench line, quick lines of every workload, kernel stats +
# one-frame timelines (default, fp16 720x960, native td4 769x1537), MFMA-busy counter pass of the default frame and of the fp16 frame
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5e; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 2 $R/build.log
echo "== -m gpu suite"
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > $R/gpu_tests.log 2>&1; tail -n 14 $R/gpu_tests.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; tail -n 2 $R/smoke.log
echo "== full default bench line"
( time timeout 900 python bench.py ) > $R/bench_default.log 2>&1; grep '^{' $R/bench_default.log | tail -1 > $R/line_default_full.json; echo "exit $?"; tail -n 4 $R/bench_default.log | cut -c1-200
echo "== full fp16 720x960 line (with PMC traffic)"
( time timeout 600 python bench.py --model td2 --backbone resnet34 --size 720x960 --precision fp16 --no-other-configs ) > $R/bench_fp16.log 2>&1; grep '^{' $R/bench_fp16.log | tail -1 > $R/line_fp16_720_full.json
echo "== quick lines"
: > $R/lines_quick.jsonl
for args in "--model td2 --size 1024x2048" "--model td2 --backbone resnet34 --size 720x960 --precision fp16" "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --clips-per-gpu 2" \
            "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --mode frame-pipelined" "--precision fp16" "--size 769x1537" "--model td2 --backbone resnet50 --size 769x1537" \
            "--model psp --backbone resnet101 --size 769x1537" "--mode frame-pipelined" "--clips-per-gpu 2"; do
  l=$(timeout 300 python bench.py --steps 60 --quick $args 2>/dev/null | grep '^{' | tail -1); echo "$l" >> $R/lines_quick.jsonl
  echo "[quick $args] $(echo "$l" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d.get('latency_ms_synced'),'ms synced', d.get('launches_per_frame'),'launches', d.get('roofline',{}).get('frac'), d.get('two_lanes_bit_identical_to_one_handle'))" 2>&1)" | tee -a $R/bench_summary.txt
done
echo "== profiles"
for cfg in "fp16_td2psp34_720x960:--model td2 --backbone resnet34 --size 720x960 --precision fp16" "td4psp18_1024x2048_default:" "td4psp18_769x1537:--size 769x1537"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$name -o r1 -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $args > $R/prof_$name.log 2>&1 )
  cp $(find $R/prof_$name -name "*kernel_stats.csv" | head -1) $R/kernel_stats_$name.csv 2>/dev/null
  python tools/timeline.py $R/prof_$name > $R/timeline_$name.txt 2>&1; head -n 1 $R/timeline_$name.txt
done
echo "== MFMA-busy counter passes"
bash tools/gpu_pmc.sh r5e_pmc > $R/pmc_default.log 2>&1; cp gpurun_out/r5e_pmc/pmc_summary.txt $R/pmc_summary_default.txt 2>/dev/null; tail -n 12 $R/pmc_summary_default.txt | cut -c1-160
bash tools/gpu_pmc.sh r5e_pmc16 --model td2 --backbone resnet34 --size 720x960 --precision fp16 > $R/pmc_fp16.log 2>&1; cp gpurun_out/r5e_pmc16/pmc_summary.txt $R/pmc_summary_fp16.txt 2>/dev/null; tail -n 12 $R/pmc_summary_fp16.txt | cut -c1-160
find $R gpurun_out/r5e_pmc gpurun_out/r5e_pmc16 -name "*kernel_trace.csv" -size +6M -delete; find $R gpurun_out/r5e_pmc gpurun_out/r5e_pmc16 -name "*.db" -delete; du -sh $R | tail -1
