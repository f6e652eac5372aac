#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_e
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "elastic or rebuild" > $O/test_rebuild.log 2>&1; tail -5 $O/test_rebuild.log
for rem in last; do
PP_BENCH_NO_COLD=1 PP_SPEC_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --remainder $rem > $O/bench_c3_$rem.json 2> $O/bench_c3_$rem.err
grep "rebuild in place" $O/bench_c3_$rem.err | cut -c1-130 | head -30
python - <<PY
import json
j=json.load(open("$O/bench_c3_$rem.json"))
print("$rem", round(j["ms_per_step"],4), j["roofline"]["phases"], j["rebuilds"])
PY
done
cd /tmp; export TMPDIR=/tmp
PP_BENCH_NO_COLD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3.csv; rm -rf $O/kt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_c3.csv")))
for r in rows[:22]:
    print("%-60s calls %5s avg %10.1f us  tot %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
