#!/bin/bash
# GPU-box probe: PMC passes over isolated launches of the fp16 conv kernels on the dominant layer4 shape (128x256x512 -> 512, 3x3 d4) --
# the LDS-DMA kernel with 256x256 / 256x128 tiles (tile codes 19 / 18) and the register-staged 128x128 kernel (3): where do the cycles go?
# Other shapes / tiles: CH_SHAPE=H,W,Cin,Cout,dil CH_TILES=a,b,c (tile codes of tdnet_op_conv2d_f16io).
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT/gpurun_out/${1:-convh_pmc}"; rm -rf $R; mkdir -p $R
cat > /tmp/ch.py <<'PY'
import sys, ctypes, os; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from tdnet_amd import _capi
lib = _capi.test_lib()
H, W, Cin, Cout, DIL = [int(v) for v in os.environ.get("CH_SHAPE", "128,256,512,512,4").split(",")]
TILES = [int(v) for v in os.environ.get("CH_TILES", "19,18,3").split(",")]
g = np.random.default_rng(0)
x = torch.from_numpy(g.standard_normal((H, W, Cin)).astype(np.float32)).cuda()
w = (g.standard_normal((Cout, Cin, 3, 3)) / 68).astype(np.float32); b = np.zeros(Cout, np.float32)
out = torch.empty(H, W, Cout, device="cuda")
for tile in TILES:
    for _ in range(4):
        lib.check(lib.tdnet_op_conv2d_f16io(x.data_ptr(), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, 3, 1, DIL, None, 1, tile, out.data_ptr(), None))
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/p$i -o r1 -- python /tmp/ch.py > $R/p$i.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - "$R" <<'PY' | tee $R/summary.txt
import csv, glob, collections, sys
R = sys.argv[1]
for i in range(1, 8):
    fs = glob.glob(R + "/p%d/**/*counter_collection.csv" % i, recursive=True)
    tr = glob.glob(R + "/p%d/**/*kernel_trace.csv" % i, recursive=True)
    if not fs or not tr:
        print("pass", i, "no counter file"); continue
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); ns = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0][:48]
        if "conv" not in k or "k_f2h" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"]); ns[k] += dur.get(r["Dispatch_Id"], 0)
    for k, v in agg.items():
        print("pass %d %-48s n=%d avg %.1f us  " % (i, k, len(n[k]), ns[k] / max(1, len(n[k])) / 1e3) + "  ".join("%s=%.4g" % (c, x / len(n[k])) for c, x in sorted(v.items())))
PY
find $R -name "*.csv" -size +2M -delete
