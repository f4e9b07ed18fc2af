#!/bin/bash
# round 6, GPU visit ak: where the cache-only attention chain forks (experiment TDNET_CHAIN_AT=<backbone block>: -1 = at the frame's start, 2 = before layer2, 4 = before layer3)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6ak; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for rep in 1 2; do
for e in -1 1 2 3 4; do
  if [ $e = -1 ]; then unset TDNET_CHAIN_AT; else export TDNET_CHAIN_AT=$e; fi
  echo "== TDNET_CHAIN_AT=$e"
  python tools/ab_opts.py --size 1024x2048 --steps 60 --rounds 2 "" "precision=2" 2>&1 | tail -2
done; done 2>&1 | tee $R/ab.txt
