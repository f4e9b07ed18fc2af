#!/bin/bash
# GPU-box visit: rocprofv3 counter passes of the default bench (one counter set per pass, never combined with sys/hip tracing):
#   pmc_sq    MFMA-pipe busy cycles, wave cycles, wait states, LDS bank conflicts   pmc_fetch / pmc_write   HBM-side bytes
# and a per-kernel summary (tools/summarize_prof.py).  tools/gpu_pmc.sh <tag> [bench args...]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pmc}; shift
R=gpurun_out/$TAG
rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick $*"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d "$GRAFT_REPO_ROOT/$R/pmc_sq" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_sq.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/$R/pmc_fetch" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/$R/pmc_write" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R > $R/pmc_summary.txt 2>&1
python - <<PY >> $R/pmc_summary.txt
import csv, glob, collections
# per-kernel: MFMA-pipe busy share and the clock the chip held (GRBM_GUI_ACTIVE cycles / kernel duration)
tr = glob.glob("$R/pmc_sq/**/*kernel_trace.csv", recursive=True); cc = glob.glob("$R/pmc_sq/**/*counter_collection.csv", recursive=True)
if tr and cc:
    dur = {}
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc[0])):
        d = dur.get(r["Dispatch_Id"])
        if not d: continue
        agg[d[1]][r["Counter_Name"]] += float(r["Counter_Value"]); agg[d[1]]["_ns_" + r["Counter_Name"]] += d[0]
    print("==== derived (pmc_sq pass).  GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:")
    print("====   clock = GRBM_GUI_ACTIVE / 8 / duration;  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("_ns_GRBM_GUI_ACTIVE", 0)):
        g, ns = v.get("GRBM_GUI_ACTIVE", 0), v.get("_ns_GRBM_GUI_ACTIVE", 0)
        if g <= 0 or ns <= 0: continue
        print("  %-60s mfma_busy %.3f   clock %.2f GHz   wait_any/wave %.2f" % (k, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * g / 8.0), g / 8.0 / ns,
              v.get("SQ_WAIT_ANY", 0) / max(1.0, v.get("SQ_WAVE_CYCLES", 0))))
PY
find $R -name "*.csv" -size +2M -delete
tail -30 $R/pmc_summary.txt
