#!/usr/bin/env python
"""Average the per-dispatch counters of a tools/pmc_run.sh output directory per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(out):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("    %-40s n=%4d mean=%16.1f" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
