#!/bin/bash
# HBM traffic of the drop-in driver's step from the PMC counters: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md), the program itself behind `--` (no launcher, no env).   tools/r05_driver_pmc.sh [iters]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_driver_pmc
mkdir -p $O
IT=${1:-20}
make -C $R/pumi-pic_amd/drivers -s || exit 1
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
c, e, cl = pp.synth.annulus_tri()
pp.synth.write_mesh_bin("/tmp/annulus100k.bin", 2, c, e, cl)
PY
cd /tmp; export TMPDIR=/tmp
D=$R/pumi-pic_amd/drivers/pseudoXGCm
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $D /tmp/annulus100k.bin 10000000 12 $IT 0.5 0 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $D /tmp/annulus100k.bin 10000000 12 $IT 0.5 0 > $O/write.log 2>&1
python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob("$O/" + sub + "/*/*counter_collection.csv") + glob.glob("$O/" + sub + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
IT = $IT
RF, WF = 1.9991, 1.0  # calibration of tools/ub_stream.hip s_rows<8,4> (profiles/traffic_c3.json): FETCH_SIZE tallies 128-B requests at 64 B on gfx950
out = {"command": "drivers/pseudoXGCm <100352-tri annulus> 10000000 12 %d 0.5 0" % IT, "kernels": {},
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); KiB x 1024 x %.4f / x %.4f" % (RF, WF)}
tot = 0.0
for name, ctr in agg.items():
    if "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
        continue
    n = min(len(ctr["FETCH_SIZE"]), len(ctr["WRITE_SIZE"]))
    if n < IT - 1:
        continue  # set-up kernels
    lps = n / float(IT)
    r = sum(ctr["FETCH_SIZE"][:n]) / IT * 1024 * RF
    w = sum(ctr["WRITE_SIZE"][:n]) / IT * 1024 * WF
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    out["kernels"][short] = {"launches_per_step": round(lps, 2), "read_bytes_per_step": r, "write_bytes_per_step": w}
    tot += r + w
out["traffic_bytes_per_step"] = tot
json.dump(out, open("$O/traffic_driver.json", "w"), indent=1)
print("traffic per step: %.1f MB" % (tot / 1e6))
for k, v in sorted(out["kernels"].items(), key=lambda kv: -(kv[1]["read_bytes_per_step"] + kv[1]["write_bytes_per_step"]))[:16]:
    print("  %-62s x%-5s R %8.1f MB  W %8.1f MB" % (k[:62], v["launches_per_step"], v["read_bytes_per_step"] / 1e6, v["write_bytes_per_step"] / 1e6))
PY
rm -rf $O/fetch $O/write
