#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_j
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ray_and_segment" 2>&1 | grep -E "passed|failed" 
cd /tmp; export TMPDIR=/tmp
export PP_BENCH_NO_COLD=1
for st in scs csr; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --workload c4 --structure $st --no-cpu-baseline --steps 40 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c4_$st.csv
t=$(find $O/kt -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" 2>&1 | head -8
rm -rf $O/kt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_c4_$st.csv")))
tot=0
for r in rows[1:40]:
    per_step=float(r["TotalDurationNs"])/1e3/43
    tot+=per_step
    if per_step>2: print("%-56s calls %5s avg %8.1f us  per-step %7.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:56], r["Calls"], float(r["AverageNs"])/1e3, per_step))
print("$st sum per step", tot)
PY
done
