#!/bin/bash
# round 3: record-fed push (deferred second pass of the re-layout): new tests, the fused-flow tests, c3 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ab_recordfed
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lazy.py -x -q > $O/pytest_lazy.txt 2>&1
tail -15 $O/pytest_lazy.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu > $O/pytest_parity.txt 2>&1
tail -5 $O/pytest_parity.txt
b() { name=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; }
b c3_lazy X=1
b c3_nolazy PP_NO_LAZY_UNPACK=1
b c3_lazy2 X=1
b c3_nolazy2 PP_NO_LAZY_UNPACK=1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 > $O/kt_c3.log 2>&1
f=$(find $O/kt_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3_lazy.csv
rm -rf $O/kt_c3
cd $R
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); print("%-28s ms/step %8.4f value %.3e frac %.3f  ps %.3f rest %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["phases"]["push_search"]["ms"], j["roofline"]["phases"]["rebuild_scatter"]["ms"]))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_c3_lazy.csv')))
for r in rows[:16]:
    print("%-60s calls %5s avg %10.1f us  min %8.1f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
