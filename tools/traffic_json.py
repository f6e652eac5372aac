#!/usr/bin/env python
"""profiles/traffic_<workload>.json from a tools/history/r01_measure.sh output directory: FETCH_SIZE and
WRITE_SIZE per kernel, scaled by the calibration run of tools/ub_stream.hip (s_rows<8,4> moves
exactly 37 B read + 32 B written per slot)."""
import collections
import csv
import glob
import json
import sys


KERNELS = {  # the kernels whose mean time bench.py reports as roofline.kernel_ms
    "c2": (("k_push_walk_rowsq<3", "k_walk_pending<3"), 69.0),
    "c3": (("k_push_walk_rowsq<3", "k_walk_pending<3"), 69.0),
    "2d": (("k_push_walk_rows<2",), 37.0),
    "c4": (("k_pseudo_push160",), 161.0),
}


def main(out_dir, workload, particles, dest, pmc_dir="pmc"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in (pmc_dir, "pmc"):  # the calibration run lives in pmc/
        for f in glob.glob(out_dir + "/" + sub + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                agg[(sub, r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if sub == "pmc" and pmc_dir == "pmc":
            break

    def mean(k, c, sub):
        for (sb, name) in agg:
            if sb == sub and k in name and c in agg[(sb, name)]:
                v = agg[(sb, name)][c]
                return sum(v) / len(v)
        return None

    kib = 1024.0
    cap = 10485760 + 64 * 8 * 100  # slots of the calibration kernel (tools/ub_stream.hip)
    rf = 37.0 * cap / (mean("s_rows<8, 4>", "FETCH_SIZE", "pmc") * kib)
    wf = 32.0 * cap / (mean("s_rows<8, 4>", "WRITE_SIZE", "pmc") * kib)
    kernels, bpp = KERNELS[workload]
    out = {"workload": workload, "particles": particles, "remainder": "last",
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py "
                     "--workload %s --steps 10 --warmup 3`, tools/history/r01_measure.sh; scaled by the calibration "
                     "run of tools/ub_stream.hip s_rows<8,4>: FETCH_SIZE[KiB] x 1024 x %.4f, "
                     "WRITE_SIZE[KiB] x 1024 x %.4f" % (workload, rf, wf),
           "kernels": {}}
    tot = 0.0
    for k in kernels:
        r = mean(k, "FETCH_SIZE", pmc_dir) * kib * rf
        w = mean(k, "WRITE_SIZE", pmc_dir) * kib * wf
        out["kernels"][k + (">" if "<" in k else "")] = {"read_bytes": r, "write_bytes": w}
        tot += r + w
    out["traffic_bytes_per_step"] = tot
    out["algorithmic_bytes_per_step"] = bpp * particles
    json.dump(out, open(dest, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], *(sys.argv[5:6]))
