#!/usr/bin/env python
"""Merge a rocprofv3 kernel trace and HIP API trace into one timeline for a step in the MIDDLE of the run (a step starts at a kernel
whose name contains the marker):   host_timeline.py kernel_trace.csv hip_api_trace.csv marker"""
import csv
import sys

kt, at, marker = sys.argv[1:4]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))))
starts = [i for i, k in enumerate(ks) if marker in k[2]]
if len(starts) < 3:
    sys.exit("fewer than three steps in the trace")
m = len(starts) // 2
i0, i1 = starts[m], starts[m + 1]
t0 = ks[i0 - 6][0] if i0 >= 6 else ks[i0][0]
t1 = ks[i1][1]
ev = []
for s, e, n in ks:
    if t0 <= s <= t1:
        ev.append((s, "GPU ", e - s, n.replace("(anonymous namespace)::", "")[:70]))
for r in csv.DictReader(open(at)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= t1:
        ev.append((s, "host", e - s, r["Function"]))
ev.sort()
for s, w, d, n in ev:
    print("%10.1f us  %s %8.1f us  %s" % ((s - t0) / 1e3, w, d / 1e3, n))
