#!/bin/bash
# round 6, GPU visit ap: the persistent grid of the fp32 k_gemm_dma for the row-parity chains' GEMMs only (experiment TDNET_DMA_CHAIN_GRID, 0 = 768), and precision 2 with the shipped 320
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6ap; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
for rep in 1 2; do
for e in 0 448 512 576 640 1024; do
  export TDNET_DMA_CHAIN_GRID=$e
  echo -n "fp32 chain grid $e: "
  python tools/ab_opts.py --size 1024x2048 --steps 60 --rounds 2 "" 2>&1 | tail -1
done; done 2>&1 | tee $R/ab.txt
unset TDNET_DMA_CHAIN_GRID
python tools/ab_opts.py --size 1024x2048 --rounds 3 "" "precision=2" 2>&1 | tail -3 | tee -a $R/ab.txt
