#!/bin/bash
# round 3, first look at the resident-record path: parity tests, then c3 / 2dc3 with records on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_resident.txt 2>&1
tail -15 $O/pytest_resident.txt
for rec in on off; do
timeout 300 python bench.py --no-cpu-baseline --records $rec > $O/bench_c3_$rec.json 2> $O/bench_c3_$rec.err
timeout 300 python bench.py --no-cpu-baseline --workload 2dc3 --records $rec > $O/bench_2dc3_$rec.json 2> $O/bench_2dc3_$rec.err
timeout 300 python bench.py --no-cpu-baseline --remainder spread --records $rec > $O/bench_c3_spread_$rec.json 2> $O/bench_c3_spread_$rec.err
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o p -- python3 $R/bench.py --no-cpu-baseline > $O/kt_c3.log 2>&1
f=$(find $O/kt_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3_rec.csv
t=$(find $O/kt_c3 -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_c3_rec.txt 2>&1
rm -rf $O/kt_c3
cd $R
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); print("%-28s ms/step %8.4f value %.3e frac %.3f  ps %.3f rest %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["phases"]["push_search"]["ms"], j["roofline"]["phases"]["rebuild_scatter"]["ms"]))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
head -25 $O/kernel_stats_c3_rec.csv | cut -c1-150
