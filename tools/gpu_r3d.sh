#!/bin/bash
# Round-3 visit D: (1) which TD_VW4_FIX variant makes the 4-channel-per-lane Winograd transforms right on the hardware, (2) fp16: the
# LDS-DMA conv kernel with 256 x 256 tiles vs without (fusion 1024) vs the register-staged kernel (fusion 128), (3) fp32 C3 A/B of the
# 4-pixel layout kernel (fusion 256) and of the channel-sliced chain attention (fusion 512 turns it off).
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3d2}
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
for m in 0 1 2 3; do echo "== TD_VW4_FIX=$m"; timeout 120 python tools/wino_vw_probe.py tools/_build/libtdnet_vw4fix$m.so 34,33 2>&1 | grep -v "^tdnet_amd\|amdgpu.ids" | cut -c1-200; done | tee $R/vw4fix_probe.txt
timeout 300 python -m pytest tests/test_gpu_fp16.py -q -m gpu -x -s > $R/gpu_fp16.log 2>&1; tail -n 5 $R/gpu_fp16.log
run() { timeout 200 python bench.py --steps 60 --quick $1 > $R/v.log 2>&1
  echo "[$1] $(tail -1 $R/v.log | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], "fps", d["ms_per_step"], "ms", d.get("breakdown_ms_per_frame"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e: print("FAILED", e)')" | tee -a $R/summary.txt; tail -4 $R/v.log | head -3 | cut -c1-300 >> $R/errs.txt; }
for f in 6 1030 134; do
  run "--model td4 --size 1024x2048 --precision fp16 --fusion $f"
  run "--model td2 --backbone resnet34 --size 720x960 --precision fp16 --fusion $f"
done
run ""; run "--fusion 262"; run "--fusion 518"; run "--fusion 774"; run ""
cd /tmp && export TMPDIR=/tmp
for cfg in "td4 resnet18 1024x2048" "td2 resnet34 720x960"; do set -- $cfg
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model $1 --backbone $2 --size $3 --precision fp16"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof_$1" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof_$1.log" 2>&1
cp $(find $GRAFT_REPO_ROOT/$R/prof_$1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$R/kernel_stats_fp16_$1_$3.csv 2>/dev/null
done
cd "$GRAFT_REPO_ROOT"
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.csv" -size +4M -delete
head -8 $R/kernel_stats_fp16_td4_1024x2048.csv | cut -c1-160
