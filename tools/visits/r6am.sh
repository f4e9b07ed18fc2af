#!/bin/bash
# round 6, GPU visit am: the cache-only chain forked later still, inside the row-parity run (experiment TDNET_CHAIN_SHIFT=k: k blocks after layer3's first)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6am; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for rep in 1 2; do
for e in 0 1 2 3; do
  export TDNET_CHAIN_SHIFT=$e
  echo "== TDNET_CHAIN_SHIFT=$e"
  python tools/ab_opts.py --size 1024x2048 --steps 60 --rounds 2 "" "precision=2" 2>&1 | tail -2
done; done 2>&1 | tee $R/ab.txt
