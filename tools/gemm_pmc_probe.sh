#!/bin/bash
# PMC on isolated persistent-GEMM launches: short K (many tile boundaries) vs long K, same FLOP per launch.
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc"; rm -rf $R; mkdir -p $R
cat > /tmp/gp.py <<'PY'
import sys, ctypes; sys.path.insert(0, "/root/repo")
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
o = lib.opts(winograd=0)
lib.tdnet_bench_conv(1024, 256, 128, 512, 1, 1, 1, 3, 6, ctypes.byref(o), None)     # M = 262144, K = 128: 16 tiles per workgroup x 4 steps
lib.tdnet_bench_conv(64, 256, 2048, 512, 1, 1, 1, 3, 6, ctypes.byref(o), None)      # M = 16384, K = 2048: 1 tile per workgroup x 64 steps (same FLOP)
lib.tdnet_bench_conv(256, 256, 512, 512, 1, 1, 1, 3, 6, ctypes.byref(o), None)      # M = 65536, K = 512: 4 tiles x 16 steps (same FLOP)
PY
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/p$i -o r1 -- python /tmp/gp.py > $R/p$i.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for i in range(1, 6):
    fs = glob.glob("gpurun_out/gemm_pmc/p%d/**/*counter_collection.csv" % i, recursive=True)
    tr = glob.glob("gpurun_out/gemm_pmc/p%d/**/*kernel_trace.csv" % i, recursive=True)
    if not fs or not tr:
        print("pass", i, "no counter file"); continue
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if "k_gemm_persistent" not in r["Kernel_Name"]: continue
        key = "grid %s" % r.get("Grid_Size", "?") + " wg " + r.get("Workgroup_Size", "?") + " ns~%d" % (round(dur.get(r["Dispatch_Id"], 0), -4))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[key]["n_" + r["Counter_Name"]] += 1
    for k, v in sorted(agg.items()):
        print("pass %d  %s" % (i, k))
        for c, x in sorted(v.items()):
            if not c.startswith("n_"): print("      %-32s %.4g  (rows %d)" % (c, x, v["n_" + c]))
PY
find $R -name "*.csv" -size +2M -delete
