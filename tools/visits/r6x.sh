#!/bin/bash
# round 6, GPU visit x: k_conv_adirect_b3 second form (fenced groups, LDS-DMA weights on three buffers): accuracy, determinism, conv time, counters, frame A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6x; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_b3.py -x -q -s -k "operators" > $R/test_ops.log 2>&1; grep -E "conv .*64->64|conv .*96->48|passed|failed|Error" $R/test_ops.log | tail -6
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $R/determinism.txt
import ctypes, numpy as np, torch
from tdnet_amd import _capi
lib = _capi.test_lib()
g = np.random.default_rng(5)
for (H, W, Cin, Cout, dil) in ((256, 512, 64, 64, 1), (193, 385, 64, 64, 1), (61, 77, 96, 48, 2)):
    x = torch.from_numpy(g.standard_normal((H, W, Cin)).astype(np.float32)).cuda()
    w = (g.standard_normal((Cout, Cin, 3, 3)) / 24).astype(np.float32)
    o = lib.opts(precision=2, fusion=lib.opts().fusion | 524288)
    outs = []
    for it in range(6):
        out = torch.full((H, W, Cout), 7e7, device="cuda")
        lib.check(lib.tdnet_op_conv2d(x.data_ptr(), H, W, Cin, w.ctypes.data, None, Cout, 3, 1, dil, None, 0, ctypes.byref(o), -1, out.data_ptr(), None))
        outs.append(out.cpu())
    print(H, W, Cin, Cout, "six runs identical:", all(torch.equal(outs[0], z) for z in outs[1:]))
PY
python tools/adb3_probe.py 256x512 2>&1 | grep -v amdgpu.ids | tee $R/conv.txt
python tools/adb3_probe.py 193x385 2>&1 | grep -v amdgpu.ids | tee -a $R/conv.txt
PMC_CMD="python $GRAFT_REPO_ROOT/tools/adb3_probe.py 256x512 10 split" bash tools/b3_pmc.sh $R/pmc_split 2>&1 | grep adirect | tee $R/pmc_split.txt
F=$(python -c "from tdnet_amd import _capi; print(_capi.lib().opts().fusion | 524288)" 2>/dev/null | tail -1)
python tools/ab_opts.py --size 1024x2048 --rounds 3 "precision=2" "precision=2,fusion=$F" 2>&1 | tail -3 | tee $R/ab_1024.txt
