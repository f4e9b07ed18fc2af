#!/bin/bash
# round 6, GPU visit ah: precision 2, narrow stride-1 1x1 convs on the split direct kernel instead of the fp32 GEMM (experiment TDNET_ADB3_1X1=1)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6ai; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for rep in 1 2; do
for e in 0 1; do
  if [ $e = 1 ]; then export TDNET_ADB3_1X1=1; else unset TDNET_ADB3_1X1; fi
  echo "== TDNET_ADB3_1X1=$e"
  python tools/ab_opts.py --model td2 --backbone resnet50 --size 769x1537 --steps 30 --rounds 2 "" "precision=2" 2>&1 | tail -2
  python tools/ab_opts.py --size 1024x2048 --steps 60 --rounds 2 "" "precision=2" 2>&1 | tail -2
done; done 2>&1 | tee $R/ab.txt
