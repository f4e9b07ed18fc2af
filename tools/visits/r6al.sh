#!/bin/bash
# round 6, GPU visit al: the cache-only chain forked in front of layer3 (fusion bit 1048576) instead of at the frame's start, across the workloads
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6al; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
F=$(python -c "from tdnet_amd import _capi; print(_capi.lib().opts().fusion | 1048576)" 2>/dev/null | tail -1)
{
python tools/ab_opts.py --size 1024x2048 --rounds 4 "" "fusion=$F" "precision=2" "precision=2,fusion=$F" 2>&1 | tail -5
python tools/ab_opts.py --size 769x1537 --rounds 4 "" "fusion=$F" "precision=2" "precision=2,fusion=$F" 2>&1 | tail -5
python tools/ab_opts.py --size 512x1024 --rounds 3 "" "fusion=$F" 2>&1 | tail -3
python tools/ab_opts.py --precision fp16 --size 1024x2048 --rounds 3 "" "fusion=$F" 2>&1 | tail -3
python tools/ab_opts.py --model td2 --backbone resnet34 --precision fp16 --size 720x960 --rounds 3 "" "fusion=$F" 2>&1 | tail -3
python tools/ab_opts.py --model td2 --backbone resnet50 --size 769x1537 --steps 30 --rounds 2 "" "fusion=$F" 2>&1 | tail -3
} | tee $R/ab.txt
