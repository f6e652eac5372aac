#!/usr/bin/env python
"""GPU idle time inside the timed region of a bench run, from a rocprofv3 --kernel-trace CSV:
sum of the gaps between consecutive kernels (same queue order) over the last N steps."""
import csv
import sys


def main(path, marker="k_push_walk_rows", last_steps=15):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(starts) < last_steps + 2:
        print("not enough steps")
        return
    a, b = starts[-last_steps - 1], starts[-1]
    busy = gap = 0
    biggest = []
    for i in range(a, b):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        busy += e - s
        g = int(rows[i + 1]["Start_Timestamp"]) - e
        if g > 0:
            gap += g
            biggest.append((g, rows[i]["Kernel_Name"][:40], rows[i + 1]["Kernel_Name"][:40]))
    n = last_steps
    print("per step: busy %.1f us, idle %.1f us (%.1f %%), launches %.1f" % (
        busy / n / 1e3, gap / n / 1e3, 100.0 * gap / (gap + busy), (b - a) / n))
    agg = {}
    for g, x, y in biggest:
        k = (x, y)
        agg[k] = agg.get(k, 0) + g
    for (x, y), g in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
        print("  %7.1f us/step between %-40s -> %s" % (g / n / 1e3, x, y))


if __name__ == "__main__":
    main(*sys.argv[1:2], **({"marker": sys.argv[2]} if len(sys.argv) > 2 else {}))
