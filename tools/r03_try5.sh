#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
run() { name=$1; envs=$2; shift 2; timeout 300 env $envs python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run c3 X=1
run c3_nofusedsort PP_NO_FUSED_SORT=1
run c3_spread X=1 --remainder spread
run 2dc3 X=1 --workload 2dc3
run 2dc3_nolazy PP_NO_LAZY_UNPACK=1 --workload 2dc3
run c4 X=1 --workload c4
run c4_nofusedsort PP_NO_FUSED_SORT=1 --workload c4
run c3_b X=1
run c3_nofusedsort_b PP_NO_FUSED_SORT=1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); ph=j["roofline"].get("phases",{}); print("%-28s ms/step %8.4f value %.3e frac %.3f  ps %.3f rest %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"], ph.get("push_search",{}).get("ms") or 0, ph.get("rebuild_scatter",{}).get("ms") or 0))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
