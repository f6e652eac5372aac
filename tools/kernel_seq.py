#!/usr/bin/env python
"""the kernels of the last full step of a rocprofv3 --kernel-trace CSV, in time order: start offset, duration,
gap to the previous one -- tools/kernel_seq.py <kernel_trace.csv> <marker kernel substring>"""
import csv
import sys


def main(path, marker):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(starts) < 6:
        print("not enough steps")
        return
    a, b = starts[-5], starts[-4]  # (not the last one: drivers print their metrics there)
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = None
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap,
                                                     r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]))
        prev_end = e
    print("step: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
