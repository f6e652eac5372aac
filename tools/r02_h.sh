#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_h
mkdir -p $O
cd $R
export PP_BENCH_NO_COLD=1
timeout 900 python bench.py --workload c5 --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/bench_c5_1m.json 2> $O/c5.err; tail -2 $O/c5.err
timeout 900 python bench.py --workload c3 --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/bench_c3_1m.json 2>/dev/null
timeout 900 python bench.py --workload c2 --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/bench_c2_1m.json 2>/dev/null
timeout 300 python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null
timeout 300 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null
python - <<PY
import json
for n in ("c5_1m","c3_1m","c2_1m","c4","c2"):
    try:
        j=json.load(open("$O/bench_%s.json"%n))
        print(n, "ms/step %.4f"%j["ms_per_step"], "value %.3e"%j["value"], "frac %.3f"%j["roofline"]["frac"], j["roofline"].get("phases",{}).get("push_search",{}).get("ms"), j.get("rebuilds"))
    except Exception as e: print(n, "FAILED", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --workload c5 --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c5_1m.csv; rm -rf $O/kt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_c5_1m.csv")))
for r in rows[1:24]:
    print("%-58s calls %5s avg %8.1f us  per-step %7.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:58], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/13))
PY
