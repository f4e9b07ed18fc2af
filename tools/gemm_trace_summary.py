#!/usr/bin/env python3
"""Summarise tools/_build/gemm_trace dumps: per co-residency group (workgroup index / 256 = wave slot on the SIMD) the number of tiles,
mean K-loop and epilogue durations and finish times."""
import collections, sys
import numpy as np
for f in sys.argv[1:]:
    lines = open(f).read().splitlines()
    print(lines[0])
    end = collections.defaultdict(list); kl = collections.defaultdict(list); ep = collections.defaultdict(list); nt = collections.defaultdict(list)
    for l in lines:
        if l.startswith("#") or "|" not in l: continue
        a, b, c = l.split("|")
        wg = int(a.split()[0]); t = list(map(float, c.split())); prev = float(b); n = 0
        for i in range(0, len(t) - 1, 2):
            if t[i] <= prev or t[i] > 1e7: break
            kl[wg >> 8].append(t[i] - prev); ep[wg >> 8].append(t[i + 1] - t[i]); prev = t[i + 1]; n += 1
        end[wg >> 8].append(prev); nt[wg >> 8].append(n)
    for k in sorted(end):
        print("  group %d: %d workgroups, %.1f tiles each | K loop mean %.1f us | epilogue mean %.1f us | finished at mean %.1f min %.1f max %.1f us"
              % (k, len(end[k]), np.mean(nt[k]), np.mean(kl[k]), np.mean(ep[k]), np.mean(end[k]), np.min(end[k]), np.max(end[k])))
