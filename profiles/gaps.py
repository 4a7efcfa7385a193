#!/usr/bin/env python3
"""Idle time between consecutive kernels of one queue, from a rocprofv3 --kernel-trace CSV (run on the GPU box, where the
raw trace lives): usage gaps.py <dir with *kernel_trace.csv>.  Prints, per (previous kernel -> next kernel) pair of the
busiest queue, the number of transitions and the mean / median gap between the end of one and the start of the other."""
import csv
import glob
import statistics
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:3]:
    ks.sort()
    gaps = defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        gaps[(n0, n1)].append(s1 - e0)
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    print("queue %s: %d kernels, busy %.1f ms of %.1f ms" % (q, len(ks), busy / 1e6, span / 1e6))
    for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:6]:
        print("   %-40s -> %-40s n=%6d  gap mean %8.2f us  median %8.2f us" % (a, b, len(g), statistics.mean(g) / 1e3, statistics.median(g) / 1e3))
