#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try16
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest.txt 2>&1
grep -n "passed\|failed" $O/pytest.txt | tail -2
PP_SCATTER_ATOMIC=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lazy.py -x -q -m gpu 2>&1 | tail -1
run() { name=$1; envs=$2; shift 2; timeout 300 env $envs PP_BENCH_NO_EXTRAS=1 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run c3_a X=1
run c3_b X=1
run 2dc3 X=1 --workload 2dc3
cd /tmp; export TMPDIR=/tmp
PP_BENCH_NO_EXTRAS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 40 > $O/kt_c3.log 2>&1
t=$(find $O/kt_c3 -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_c3.txt 2>&1; rm -rf $O/kt_c3
cd $R
head -2 $O/gaps_c3.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); ph=j["roofline"].get("phases",{}); print("%-20s ms/step %8.4f frac %.3f  ps %.3f rest %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["roofline"]["frac"], ph.get("push_search",{}).get("ms") or 0, ph.get("rebuild_scatter",{}).get("ms") or 0))
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
