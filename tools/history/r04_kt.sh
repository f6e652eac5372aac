#!/bin/bash
# kernel stats of one bench.py variant: tools/r04_kt.sh <name> [bench args...]; env passes through
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_kt
mkdir -p $O
name=$1; shift
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1 PP_BENCH_NO_COLD=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/kt_$name.log 2>&1
f=$(find $O/kt_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv
t=$(find $O/kt_$name -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_$name.txt 2>&1
rm -rf $O/kt_$name
echo "== $name"
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats_$name.csv")))[:${TOPN:-12}]:
    if "closest_point" in r["Name"] or "k_inv_" in r["Name"]: continue
    print("%-72s calls %5s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:72], r["Calls"], float(r["AverageNs"])/1e3))
PY
grep "per step" $O/gaps_$name.txt
