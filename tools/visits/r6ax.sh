#!/bin/bash
# round 6, GPU visit ax: the 128-column tile of the split direct kernel (NT = 4) for convs of 65..128 output channels against two 64-column tiles (TDNET_ADB3_NT2=1)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6ax; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_b3.py -x -q -k "operators or bottleneck" > $R/tests.log 2>&1; tail -2 $R/tests.log
{
python tools/env_ab.py TDNET_ADB3_NT2 1024x2048 2>&1 | tail -1
for rep in 1 2; do for e in 0 1; do
  if [ $e = 1 ]; then export TDNET_ADB3_NT2=1; else unset TDNET_ADB3_NT2; fi
  echo -n "two 64-column tiles=$e: td2-psp50 "; python bench.py --steps 40 --quick --model td2 --backbone resnet50 --size 769x1537 --precision bf16x3 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])'
  echo -n "two 64-column tiles=$e: psp101 "; python bench.py --steps 40 --quick --model psp --backbone resnet101 --size 769x1537 --precision bf16x3 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])'
done; done
} 2>&1 | tee $R/ab.txt
