#!/bin/bash
# round 6, GPU visit af: the one-wave-per-tile transform kernels (k_wino4_in_c / out_c<4>) for EVERY Winograd conv (overlap bit 2), not only the row-parity chunks
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6af; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
python tools/ab_opts.py --size 1024x2048 --rounds 3 "" "overlap=43" "precision=2" "precision=2,overlap=43" 2>&1 | tail -5 | tee $R/ab_1024.txt
python tools/ab_opts.py --size 769x1537 --rounds 3 "" "overlap=43" "precision=2" "precision=2,overlap=43" 2>&1 | tail -5 | tee $R/ab_769.txt
