#!/bin/bash
# duration of every push / pending-walk launch of a c2 run, in launch order (config 2 never rebuilds: the particles
# drift away from their rows' elements, the launches get slower): tools/r04_c2_drift.sh [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_c2drift
mkdir -p $O
S=${1:-20}
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1 PP_BENCH_NO_COLD=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python3 $R/bench.py --workload c2 --no-cpu-baseline --steps $S --warmup 3 > $O/log.txt 2>&1
t=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$t")) if "k_push_walk_rowsq" in r["Kernel_Name"] or "k_walk_pending" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]
for r in rows:
    out.append(("push" if "rowsq" in r["Kernel_Name"] else "pend", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
push=[d for k,d in out if k=="push"]; pend=[d for k,d in out if k=="pend"]
print("push us:", " ".join("%.0f"%d for d in push))
print("pend us:", " ".join("%.0f"%d for d in pend))
PY
rm -rf $O/kt
tail -1 $O/log.txt | cut -c1-300
