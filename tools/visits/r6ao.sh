#!/bin/bash
# round 6, GPU visit ao: the persistent grid of k_gemm_b3 for the row-parity CHAINS' GEMMs only (two in flight): experiment TDNET_B3_CHAIN_GRID
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6ao; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for rep in 1 2; do
for e in 0 288 320 352 384 448; do
  export TDNET_B3_CHAIN_GRID=$e
  echo -n "chain grid $e: "
  python tools/ab_opts.py --size 1024x2048 --steps 60 --rounds 2 "precision=2" 2>&1 | tail -1
done; done 2>&1 | tee $R/ab.txt
