#!/bin/bash
# round 3: resident records, full re-layout and in-place form: parity tests, then c3 / 2dc3 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_resident.txt 2>&1
tail -25 $O/pytest_resident.txt
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b c3_soa_ref --records off
b c3_rec_ref --records on
b c3_soa_inplace --records off --rebuild in-place
b c3_rec_inplace --records on --rebuild in-place
b 2dc3_rec_inplace --workload 2dc3 --records on --rebuild in-place
b c3_spread_rec_inplace --remainder spread --records on --rebuild in-place
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o p -- python3 $R/bench.py --no-cpu-baseline --rebuild in-place --steps 40 > $O/kt_c3.log 2>&1
f=$(find $O/kt_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3_rec_inplace.csv
rm -rf $O/kt_c3
cd $R
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); print("%-28s ms/step %8.4f value %.3e frac %.3f  ps %.3f rest %.3f  %s" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["phases"]["push_search"]["ms"], j["roofline"]["phases"]["rebuild_scatter"]["ms"], j.get("rebuilds")))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_c3_rec_inplace.csv')))
for r in rows[:30]:
    print("%-60s calls %5s avg %10.1f us  min %8.1f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
