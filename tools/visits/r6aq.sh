#!/bin/bash
# round 6, GPU visit aq: do the row-parity chains pay on smaller maps with precision 2? (overlap 45 = 41 | 4 forces them below the 24000-pixel rule)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6aq; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
{
python tools/ab_opts.py --size 769x1537 --rounds 3 "precision=2" "precision=2,overlap=45" 2>&1 | tail -3
python tools/ab_opts.py --size 640x1280 --rounds 3 "precision=2" "precision=2,overlap=45" 2>&1 | tail -3
python tools/ab_opts.py --size 896x1792 --rounds 3 "precision=2" "precision=2,overlap=40" 2>&1 | tail -3
python tools/ab_opts.py --size 1024x2048 --rounds 3 "precision=2" "precision=2,overlap=40" 2>&1 | tail -3
} | tee $R/ab.txt
