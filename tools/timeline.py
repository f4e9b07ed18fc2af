#!/usr/bin/env python3
"""Print the kernel timeline of ONE steady-state frame from a rocprofv3 kernel trace (kernel_trace.csv): start offset, duration,
queue, short kernel name -- and, per queue pair, how much of the frame had kernels of BOTH queues in flight (the overlap the
row-parity chains and the side-stream attention chain are there to create).
    tools/timeline.py <dir with *kernel_trace.csv> [frame index from the end, default 2]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "").split("(")[0][:64]))
rows.sort()
# a frame starts with the layout kernel of the stem
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_nchw3_to_")]   # the frame's first kernel: NHWC4 or packed-row image
if len(starts) < back + 1:
    sys.exit("not enough frames in the trace")
a, b = starts[-back - 1], starts[-back]
fr = rows[a:b]
t0 = fr[0][0]
end = max(r[1] for r in fr)
print("frame of %d kernels, %.1f us from first start to last end" % (len(fr), (end - t0) / 1e3))
queues = sorted(set(r[2] for r in fr))
for s, e, q, name in fr:
    print("%9.1f +%8.1f  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, "    " * queues.index(q), name))
# busy time per queue and pairwise overlap (union of intervals per queue, then intersections)
def union(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: t += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
u = {q: union([(s, e) for s, e, qq, _ in fr if qq == q]) for q in queues}
for q in queues:
    print("queue %s busy %.1f us" % (q, sum(e - s for s, e in u[q]) / 1e3))
for i, q in enumerate(queues):
    for r in queues[i + 1:]:
        print("queues %s & %s both busy %.1f us" % (q, r, inter(u[q], u[r]) / 1e3))
