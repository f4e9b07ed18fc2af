#!/bin/bash
# GPU-box probe: isolated fp16 conv launches by tile code of tdnet_op_conv2d_f16io, kernel durations from a rocprofv3 kernel trace (launches
# matched by order; 6 launches per case, the last 4 averaged).   tools/conv_h_probe.sh <tag> '<python list of cases>'
#   case = (label, H, W, Cin, Cout, dil, (tile codes...), factor)     factor: the real conv's FLOP = this launch's FLOP x factor (emulations)
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT/gpurun_out/${1:-convh}"; rm -rf $R; mkdir -p $R
export RP_CASES="$2"
cat > /tmp/rp.py <<'PY'
import sys, os, json; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from tdnet_amd import _capi
lib = _capi.test_lib()
g = np.random.default_rng(0)
cases = eval(os.environ["RP_CASES"])
order = []
for name, H, W, Cin, Cout, d, tiles, fac in cases:
    x = torch.from_numpy(g.standard_normal((H, W, Cin)).astype(np.float32)).cuda()
    w = (g.standard_normal((Cout, Cin, 3, 3)) / (3 * Cin ** 0.5)).astype(np.float32); b = np.zeros(Cout, np.float32)
    out = torch.empty(H, W, Cout, device="cuda")
    for tile in tiles:
        for _ in range(6):
            lib.check(lib.tdnet_op_conv2d_f16io(x.data_ptr(), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, 3, 1, d, None, 1, tile, out.data_ptr(), None))
        order.append([name, tile, 2.0 * H * W * Cin * Cout * 9, fac])
torch.cuda.synchronize()
json.dump(order, open(os.environ["RP_ORDER"], "w"))
PY
cd /tmp && export TMPDIR=/tmp RP_ORDER=$R/order.json
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $R/p -o r1 -- python /tmp/rp.py > $R/p.log 2>&1 || tail -5 $R/p.log
cd "$GRAFT_REPO_ROOT"
python - "$R" <<'PY' | tee $R/summary.txt
import csv, glob, json, sys
R = sys.argv[1]
tr = glob.glob(R + "/p/**/*kernel_trace.csv", recursive=True)
order = json.load(open(R + "/order.json"))
rows = [r for r in csv.DictReader(open(tr[0])) if "k_conv_dma_" in r["Kernel_Name"] or "k_conv_igemm_h" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == 6 * len(order), (len(rows), len(order))
for i, (name, tile, flop, fac) in enumerate(order):
    grp = rows[6 * i + 2: 6 * i + 6]
    us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / len(grp) / 1e3
    k = grp[0]["Kernel_Name"].replace("void ", "").split("(")[0]
    print("%-36s tile %2d  %-46s %7.1f us%s  %6.0f TFLOP/s  %.3f of 2500" % (name, tile, k, us, "" if fac == 1.0 else " (x %.2f = %.1f)" % (fac, us * fac), flop / us / 1e6, flop / us / 1e6 / 2500))
PY
find $R -name "*.csv" -size +2M -delete
