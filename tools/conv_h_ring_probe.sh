#!/bin/bash
# GPU-box probe: the fp16 LDS-DMA conv at the shapes whose grid does not fill the chip (720x960: 10800 pixels; 1024x2048 layer2: 256
# workgroups) and the dominant ones -- tile codes of tdnet_op_conv2d_f16io (16 .. 22; + 32 = tap-by-tap staging instead of row images).
# Kernel durations from a rocprofv3 kernel trace, launches matched by order.
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT/gpurun_out/${1:-ring}"; rm -rf $R; mkdir -p $R
cat > /tmp/rp.py <<'PY'
import sys, os, json; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from tdnet_amd import _capi
lib = _capi.test_lib()
g = np.random.default_rng(0)
cases = [("1024x2048 layer1 64ch d1", 256, 512, 64, 64, 1, (2, 30)),
         ("720x960 layer1 64ch d1", 180, 240, 64, 64, 1, (2, 30)),
         ("769x1537 layer1 64ch d1", 193, 385, 64, 64, 1, (2, 30)),
         ("1024x2048 layer4 512ch d4", 128, 256, 512, 512, 4, (19,))]
order = []
for name, H, W, Cin, Cout, d, tiles in cases:
    x = torch.from_numpy(g.standard_normal((H, W, Cin)).astype(np.float32)).cuda()
    w = (g.standard_normal((Cout, Cin, 3, 3)) / (3 * Cin ** 0.5)).astype(np.float32); b = np.zeros(Cout, np.float32)
    out = torch.empty(H, W, Cout, device="cuda")
    res = torch.randn(H, W, Cout, device="cuda")
    for with_res in ((False, True) if os.environ.get("RP_RESID") else (False,)):
        for tile in tiles:
            for _ in range(6):
                lib.check(lib.tdnet_op_conv2d_f16io(x.data_ptr(), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, 3, 1, d, res.data_ptr() if with_res else None, 1, tile, out.data_ptr(), None))
            order.append([name + (" +resid" if with_res else ""), tile, 2.0 * H * W * Cin * Cout * 9])
torch.cuda.synchronize()
json.dump(order, open(os.environ["RP_ORDER"], "w"))
PY
cd /tmp && export TMPDIR=/tmp RP_ORDER=$R/order.json
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/p -o r1 -- python /tmp/rp.py > $R/p.log 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$R" <<'PY' | tee $R/summary.txt
import csv, glob, json, sys
R = sys.argv[1]
tr = glob.glob(R + "/p/**/*kernel_trace.csv", recursive=True)
order = json.load(open(R + "/order.json"))
rows = [r for r in csv.DictReader(open(tr[0])) if "k_conv_dma_" in r["Kernel_Name"] or "k_conv_igemm_h" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == 6 * len(order), (len(rows), len(order))
for i, (name, tile, flop) in enumerate(order):
    grp = rows[6 * i + 2: 6 * i + 6]
    us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / len(grp) / 1e3
    k = grp[0]["Kernel_Name"].replace("void ", "").split("(")[0]
    print("%-34s tile %2d  %-52s %7.1f us  %6.0f TFLOP/s  %.3f of 2500" % (name, tile, k, us, flop / us / 1e6, flop / us / 1e6 / 2500))
PY
find $R -name "*.csv" -size +2M -delete
