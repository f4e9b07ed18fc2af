#!/bin/bash
# PMC comparison of the persistent GEMM (K = 2048) and the direct 3x3 conv kernel on isolated launches.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT/gpurun_out"
cat > /tmp/gp.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch
from tdnet_amd import _capi
lib = _capi.lib(); torch.zeros(1, device="cuda")
lib.tdnet_set_conv_winograd(0)
lib.tdnet_bench_conv(128, 256, 2048, 512, 1, 1, 1, 3, 10, None)
lib.tdnet_bench_conv(128, 256, 512, 512, 3, 1, 1, 3, 5, None)
PY
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rm -rf $R/gpmc$i
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/gpmc$i -o r1 -- python /tmp/gp.py > $R/gpmc$i.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for i in (1, 2, 3):
    fs = glob.glob("gpurun_out/gpmc%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("pass", i, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print("pass %d  %s" % (i, k))
        for c, x in sorted(v.items()):
            print("      %-32s %.4g" % (c, x))
PY
find $R -name "*.csv" -size +8M -delete
