#!/bin/bash
# round 6, GPU visit v: precision 2 on ResNet layer1's convs (k_conv_adirect_b3, fusion bit 524288): accuracy, conv time, frame A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6v; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_b3.py -x -q -s -k "operators" > $R/test_ops.log 2>&1; grep -E "conv |passed|failed|Error" $R/test_ops.log | tail -14
python tools/adb3_probe.py 256x512 2>&1 | grep -v amdgpu.ids | tee $R/conv.txt
python tools/adb3_probe.py 193x385 2>&1 | grep -v amdgpu.ids | tee -a $R/conv.txt
F=$(python -c "from tdnet_amd import _capi; print(_capi.lib().opts().fusion | 524288)" 2>/dev/null | tail -1)
python tools/ab_opts.py --size 1024x2048 --rounds 3 "precision=2" "precision=2,fusion=$F" 2>&1 | tail -6 | tee $R/ab_1024.txt
python tools/ab_opts.py --size 769x1537 --rounds 3 "precision=2" "precision=2,fusion=$F" 2>&1 | tail -6 | tee $R/ab_769.txt
python tools/ab_opts.py --model td2 --backbone resnet34 --size 720x960 --rounds 2 "precision=2" "precision=2,fusion=$F" 2>&1 | tail -6 | tee $R/ab_td2.txt
