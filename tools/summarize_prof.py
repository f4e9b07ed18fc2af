#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) per kernel name: count, total/avg duration, counter sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(n):
    return n.replace("void ", "").split("(")[0][:70]


for sub in ("prof", "pmc_sq", "pmc_fetch", "pmc_write"):
    d = os.path.join(root, sub)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cfiles = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    print("==== %s: %s %s" % (sub, [os.path.basename(f) for f in files], [os.path.basename(f) for f in cfiles]))
    if files:
        agg = defaultdict(lambda: [0, 0.0])
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot = sum(v[1] for v in agg.values())
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("  %-72s n=%5d total=%10.1f us (%5.1f%%) avg=%9.2f us" % (k, v[0], v[1], 100 * v[1] / tot, v[1] / v[0]))
    if cfiles:
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        with open(cfiles[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(k, r["Counter_Name"])] += 1
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            print("  %-72s " % k + "  ".join("%s=%.4g (n=%d)" % (c, x, cnt[(k, c)]) for c, x in sorted(v.items())))
