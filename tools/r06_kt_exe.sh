#!/bin/bash
# kernel stats of any executable: tools/r06_kt_exe.sh <name> <exe> [args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_kt
mkdir -p $O
name=$1; shift
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o p -- "$@" > $O/kt_$name.log 2>&1
f=$(find $O/kt_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv
rm -rf $O/kt_$name
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats_$name.csv")))[:${TOPN:-14}]:
    print("%-84s calls %5s avg %9.1f us total %9.1f ms" % (r["Name"].replace("(anonymous namespace)::","")[:84], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
