#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_g
mkdir -p $O
cd $R
PP_BENCH_NO_COLD=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
j=json.load(open("$O/bench_c3.json"))
print("c3", round(j["ms_per_step"],4), j["roofline"]["phases"]["push_search"]["ms"], j["roofline"]["phases"]["rebuild_scatter"]["ms"], j["rebuilds"])
PY
cd /tmp; export TMPDIR=/tmp
PP_BENCH_NO_COLD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3.csv
t=$(find $O/kt -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps.txt 2>&1; tail -15 $O/gaps.txt
rm -rf $O/kt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_c3.csv")))
tot=0
for r in rows[1:40]:
    per_step=float(r["TotalDurationNs"])/1e3/43
    tot+=per_step
    print("%-58s calls %5s avg %8.1f us  per-step %7.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:58], r["Calls"], float(r["AverageNs"])/1e3, per_step))
print("sum per step", tot)
PY
