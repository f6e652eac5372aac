#!/usr/bin/env python
"""profiles/traffic_<workload>.json for a WHOLE step (c3 / c5): FETCH_SIZE and WRITE_SIZE of every kernel
a step launches, summed, per step -- from the two --pmc passes of tools/history/r04_measure.sh (round 3: r03_measure.sh) (`bench.py
--workload W --steps K --warmup W0` with PP_BENCH_NO_COLD=1 PP_BENCH_PREWARM=0, so every per-step
kernel runs exactly K + W0 times) and scaled by the calibration run of tools/ub_stream.hip (s_rows<8,4>
moves exactly 37 B read + 32 B written per slot; MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests
at 64 B on gfx950).  Kernels that run fewer than K + W0 times (set-up: ring maps, map transposes) are
left out."""
import collections
import csv
import glob
import json
import sys


def main(out_dir, workload, particles, nsteps, dest, pmc_dir, bytes_per_particle):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in (pmc_dir, "pmc"):
        for f in glob.glob(out_dir + "/" + sub + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                agg[(sub, r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))

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
    out = {"workload": workload, "particles": particles, "remainder": "last", "steps_profiled": nsteps,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --workload %s "
                     "--steps 10 --warmup 3 --no-cpu-baseline` (PP_BENCH_NO_COLD=1 PP_BENCH_PREWARM=0), "
                     "tools/r06_measure.sh (rounds 3-5: r03_ / r04_ / r05_measure.sh); every kernel that runs once per step or more, summed per step; scaled by "
                     "the calibration run of tools/ub_stream.hip s_rows<8,4>: FETCH_SIZE[KiB] x 1024 x %.4f, "
                     "WRITE_SIZE[KiB] x 1024 x %.4f" % (workload, rf, wf),
           "kernels": {}}
    tot = 0.0
    for (sub, name), ctr in sorted(agg.items()):
        if sub != pmc_dir or "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
            continue
        n = min(len(ctr["FETCH_SIZE"]), len(ctr["WRITE_SIZE"]))
        if n < nsteps - 1 or "k_closest_point" in name or name.startswith("__amd_rocclr"):
            continue  # (runtime fills / copies: < 1 MB per step; their dispatch list also holds the set-up memsets)
        # per launch x launches per step: the record-fed push runs in every step but the first (whose push
        # reads the SoA arrays the structure was built into), so it has nsteps - 1 launches
        lps = max(1, round(n / nsteps))
        r = sum(ctr["FETCH_SIZE"][:n]) / n * lps * kib * rf
        w = sum(ctr["WRITE_SIZE"][:n]) / n * lps * kib * wf
        short = name.replace("(anonymous namespace)::", "").split("(")[0]
        out["kernels"][short] = {"launches_per_step": round(n / nsteps, 2), "read_bytes_per_step": r,
                                 "write_bytes_per_step": w}
        tot += r + w
    out["traffic_bytes_per_step"] = tot
    out["algorithmic_bytes_per_step"] = bytes_per_particle * particles
    json.dump(out, open(dest, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -(kv[1]["read_bytes_per_step"] + kv[1]["write_bytes_per_step"])):
        print("  %-40s x%-5s R %8.1f MB  W %8.1f MB" % (k[:40], v["launches_per_step"], v["read_bytes_per_step"] / 1e6,
                                                        v["write_bytes_per_step"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], float(sys.argv[7]))
