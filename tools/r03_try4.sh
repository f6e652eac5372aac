#!/bin/bash
# round 3: fused radix passes + 2-D record-fed push: tests, then c3 / 2dc3 / c4 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lazy.py -x -q > $O/pytest_lazy.txt 2>&1
tail -15 $O/pytest_lazy.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu > $O/pytest_parity.txt 2>&1
tail -5 $O/pytest_parity.txt
b() { name=$1; shift; env "$@" > /dev/null; }
run() { name=$1; envs=$2; shift 2; timeout 300 env $envs python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run c3 X=1
run c3_nofusedsort PP_NO_FUSED_SORT=1
run c3_nolazy PP_NO_LAZY_UNPACK=1
run c3_spread X=1 --remainder spread
run 2dc3 X=1 --workload 2dc3
run 2dc3_nolazy PP_NO_LAZY_UNPACK=1 --workload 2dc3
run c4 X=1 --workload c4
run c4_nofusedsort PP_NO_FUSED_SORT=1 --workload c4
run c3_b X=1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 > $O/kt_c3.log 2>&1
f=$(find $O/kt_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3.csv
t=$(find $O/kt_c3 -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_c3.txt 2>&1
rm -rf $O/kt_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2dc3 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 --workload 2dc3 > $O/kt_2dc3.log 2>&1
f=$(find $O/kt_2dc3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_2dc3.csv
rm -rf $O/kt_2dc3
cd $R
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); ph=j["roofline"].get("phases",{}); print("%-28s ms/step %8.4f value %.3e frac %.3f  ps %.3f rest %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"], ph.get("push_search",{}).get("ms") or 0, ph.get("rebuild_scatter",{}).get("ms") or 0))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
python - <<PY
import csv
for n in ("c3","2dc3"):
    rows=list(csv.DictReader(open('$O/kernel_stats_%s.csv'%n)))
    print(n)
    for r in rows[:18]:
        print("  %-60s calls %5s avg %10.1f us  min %8.1f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
cat $O/gaps_c3.txt | head -12
