#!/bin/bash
# GPU-box visit: persistent GEMM A/B + parity suite.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_winograd.py -q -m gpu -x 2>&1 | tail -8 > $R/ops.log
timeout 300 python tools/wino_probe.py > $R/wino_probe_persist.log 2>&1
python - > $R/gemm_ab.log 2>&1 <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from tdnet_amd import _capi
lib = _capi.lib(); torch.zeros(1, device="cuda")
shapes = [("enc_v 1x1 512->512 @128x256", 128, 256, 512, 512), ("r50 l3 1x1 1024->256 @97x193", 97, 193, 1024, 256), ("r50 l4 1x1 512->2048 @97x193", 97, 193, 512, 2048),
          ("r50 l1 1x1 64->256 @193x385", 193, 385, 64, 256), ("ds 1x1 256->512 @128x256", 128, 256, 256, 512)]
for nm, H, W, Cin, Cout in shapes:
    gf = 2.0 * H * W * Cin * Cout / 1e9
    row = []
    for mode in (0, 1):
        lib.tdnet_set_gemm_persistent(mode)
        for t in (3, 4):
            ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 1, 1, 1, t, 20, None) for _ in range(2))
            row.append("%s %s %.3f ms %.0f TF" % ("persist" if mode else "1-tile ", "128x128" if t == 3 else "64x128", ms, gf / ms))
    print("%-32s %5.1f GF | " % (nm, gf) + " | ".join(row), flush=True)
lib.tdnet_set_gemm_persistent(1)
PY
for m in 0 1; do python - > $R/bench_gemm$m.log 2>&1 <<PY
import sys; sys.path.insert(0, ".")
from tdnet_amd import _capi; _capi.lib().tdnet_set_gemm_persistent($m)
import runpy; sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "40"]; runpy.run_path("bench.py", run_name="__main__")
PY
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_psp101.py -q -m gpu -s 2>&1 | tail -20 > $R/model.log
timeout 300 python bench.py --model td2 --backbone resnet50 --size 769x1537 --steps 40 --no-cpu-baseline > $R/bench_td2psp50.log 2>&1
tail -4 $R/ops.log $R/model.log; cat $R/gemm_ab.log; tail -10 $R/wino_probe_persist.log
