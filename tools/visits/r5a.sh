#!/bin/bash
# round 5, GPU visit a: fp16 row-parity chains A/B (bit-identity + frames/s), layer-4 mod-4 classes A/B, Infinity-Cache probe, profiles
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5a; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 2 $R/build.log
echo "== A/B fp16 td2-psp34 720x960 (first variant = unchained reference)"
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --model td2 --backbone resnet34 --size 720x960 --precision fp16 --steps 80 --rounds 3 \
    "overlap=0" "" "chain_rows=2" "chain_rows=3" "chain_rows=4" "overlap=0,fusion=8230" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp16_720.txt
echo "== A/B fp32 td4-psp18 1024x2048"
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --steps 60 --rounds 3 "" "overlap=105" "overlap=0" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp32_c3.txt
echo "== A/B fp16 td4-psp18 1024x2048 (chains on large maps: overlap bit 4)"
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --precision fp16 --steps 60 --rounds 2 "overlap=0" "overlap=45" "overlap=45,chain_rows=4" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp16_c3.txt
echo "== Infinity-Cache probe"
timeout 300 python tools/wino_l3_probe.py 2>&1 | grep -v amdgpu.ids | tee $R/l3_probe.txt
for v in whole half quarter; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/l3_$v -o p -- python $GRAFT_REPO_ROOT/tools/wino_l3_probe.py --variant $v > $R/l3_$v.log 2>&1 )
  f=$(find $R/l3_$v -name "*kernel_stats.csv" | head -1); echo "-- $v"; head -n 5 "$f" | cut -c1-140; cp "$f" $R/l3_kernel_stats_$v.csv 2>/dev/null
done
echo "== profile: fp16 720x960 frame (kernel stats + one-frame timeline)"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_fp16 -o r1 -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 > $R/prof_fp16.log 2>&1 )
cp $(find $R/prof_fp16 -name "*kernel_stats.csv" | head -1) $R/kernel_stats_fp16_720.csv 2>/dev/null
python tools/timeline.py $R/prof_fp16 > $R/timeline_fp16_720.txt 2>&1; head -n 3 $R/timeline_fp16_720.txt; tail -n 4 $R/timeline_fp16_720.txt
echo "== quick bench lines"
timeout 300 python bench.py --steps 60 --quick --model td2 --backbone resnet34 --size 720x960 --precision fp16 2>&1 | grep '^{' | tail -1 > $R/line_fp16_720.json
timeout 300 python bench.py --steps 60 --quick 2>&1 | grep '^{' | tail -1 > $R/line_default.json
python - <<'PY'
import json,os
R=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5a"
for n in ("line_fp16_720.json","line_default.json"):
    try:
        d=json.load(open(R+"/"+n)); print(n, d["value"], "fps", d.get("latency_ms_synced"), "ms synced", d.get("launches_per_frame"), "launches", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("frac_round4_kernel_set"), d.get("memory"))
    except Exception as e: print(n, "FAILED", e)
PY
find $R -name "*kernel_trace.csv" -size +6M -delete; find $R -name "*.db" -delete; du -sh $R | tail -1
