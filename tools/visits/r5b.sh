#!/bin/bash
# round 5, GPU visit b: full -m gpu suite, full default bench line (new legs), fp16 A/B of fusion bit 1, profiles of both frames
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5b; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 2 $R/build.log
echo "== A/B fp16 td2-psp34 720x960"
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --backbone resnet34 --size 720x960 --precision fp16 --steps 80 --rounds 3 \
    "" "fusion=40999" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp16_720.txt
echo "== -m gpu suite"
( time timeout 1500 python -m pytest tests -q -m gpu --durations=10 ) > $R/gpu_tests.log 2>&1; tail -n 18 $R/gpu_tests.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; tail -n 2 $R/smoke.log
echo "== full default bench line"
( time timeout 900 python bench.py ) > $R/bench_default.log 2>&1; grep '^{' $R/bench_default.log | tail -1 > $R/line_default_full.json; tail -n 4 $R/bench_default.log | cut -c1-300
echo "== profiles"
for cfg in "fp16_720:--model td2 --backbone resnet34 --size 720x960 --precision fp16" "default:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$name -o r1 -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $args > $R/prof_$name.log 2>&1 )
  cp $(find $R/prof_$name -name "*kernel_stats.csv" | head -1) $R/kernel_stats_$name.csv 2>/dev/null
  python tools/timeline.py $R/prof_$name > $R/timeline_$name.txt 2>&1; head -n 1 $R/timeline_$name.txt; tail -n 4 $R/timeline_$name.txt
done
echo "== quick lines"
timeout 300 python bench.py --steps 60 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 2>&1 | grep '^{' | tail -1 > $R/line_fp16_720.json
timeout 300 python bench.py --steps 60 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 --mode frame-pipelined 2>&1 | grep '^{' | tail -1 > $R/line_fp16_720_pipelined.json
timeout 300 python bench.py --steps 60 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 --clips-per-gpu 2 2>&1 | grep '^{' | tail -1 > $R/line_fp16_720_two_clips.json
python - <<'PY'
import json,os
R=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5b"
for n in ("line_default_full.json","line_fp16_720.json","line_fp16_720_pipelined.json","line_fp16_720_two_clips.json"):
    try:
        d=json.load(open(R+"/"+n)); print(n, d["value"], "fps", d.get("latency_ms_synced"), "ms synced", d.get("launches_per_frame"), "launches", d.get("roofline",{}).get("frac"), d.get("two_lanes_bit_identical_to_one_handle"), d.get("memory",{}).get("handle_bytes"))
        for l in d.get("other_configs",[]):
            print("   ", l["config"], l["workload"], l["value"], "fps", l.get("latency_ms_synced"), "ms", l.get("launches_per_frame"), l.get("roofline",{}).get("frac"), l.get("parity",{}).get("max_abs_dlogit"), l.get("parity",{}).get("FAILED"), l.get("published",{}).get("speedup_of_latency_ms_synced"), l.get("two_frames_in_flight",{}).get("value"), l.get("two_frames_in_flight",{}).get("second_lane"))
    except Exception as e: print(n, "FAILED", e)
PY
find $R -name "*kernel_trace.csv" -size +6M -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
