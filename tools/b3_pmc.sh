#!/bin/bash
# GPU-box: counter passes (one set per rocprofv3 run, kernel trace only) over tools/b3_one.py (or the command in $PMC_CMD); prints per-kernel derived figures.
#   tools/b3_pmc.sh <outdir> <b3_one.py args...>        PMC_CMD="python tools/adb3_probe.py 256x512 20 split" tools/b3_pmc.sh <outdir>
# (no TCC_* counters: that pass hung until its timeout on this pool, three times -- round 6)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$1; shift
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B=${PMC_CMD:-"python $GRAFT_REPO_ROOT/tools/b3_one.py $*"}
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d "$R/p1" -o r1 -- $B > "$R/p1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE -d "$R/p2" -o r1 -- $B > "$R/p2.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d "$R/p3" -o r1 -- $B > "$R/p3.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$R/p4" -o r1 -- $B > "$R/p4.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$R/p5" -o r1 -- $B > "$R/p5.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$R" <<'PY'
import csv, glob, collections, sys
R = sys.argv[1]
for p in ("p1", "p2", "p3", "p4", "p5"):
    tr = glob.glob("%s/%s/**/*kernel_trace.csv" % (R, p), recursive=True); cc = glob.glob("%s/%s/**/*counter_collection.csv" % (R, p), recursive=True)
    if not (tr and cc):
        print(p, "no output:", open("%s/%s.log" % (R, p)).read()[-400:]); continue
    dur = {}
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:48])
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); ns = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        d = dur.get(r["Dispatch_Id"])
        if not d: continue
        agg[d[1]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[d[1]] += 1; ns[d[1]] += d[0]
    for k in sorted(agg, key=lambda k: -ns[k])[:4]:
        n = cnt[k]
        print("%s %-44s n=%d avg %.1f us | " % (p, k, n, ns[k] / n / 1e3) + "  ".join("%s %.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
PY
find $R -name "*.csv" -size +1M -delete
