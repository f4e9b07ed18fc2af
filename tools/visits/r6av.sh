#!/bin/bash
# round 6, GPU visit av: the split GEMM's size rule for deep-K 1x1 convs (1024 -> 256 on 18721 rows: 148 tiles) -- experiment TDNET_B3_DEEPK=<tiles needed when K >= 1024>
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6av; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for rep in 1 2; do for e in 0 128 64; do
  export TDNET_B3_DEEPK=$e
  echo -n "deep-K rule $e: psp101 "; python bench.py --steps 40 --quick --model psp --backbone resnet101 --size 769x1537 --precision bf16x3 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])'
  echo -n "deep-K rule $e: td2-psp50 "; python bench.py --steps 40 --quick --model td2 --backbone resnet50 --size 769x1537 --precision bf16x3 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"])'
done; done 2>&1 | tee $R/ab.txt
