#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try9
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
run() { name=$1; envs=$2; shift 2; timeout 600 env $envs python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run c5_1m X=1 --workload c5 --mesh 1m --particles 32000000 --steps 10
run c5_1m_nolazy PP_NO_LAZY_UNPACK=1 --workload c5 --mesh 1m --particles 32000000 --steps 10
run c2 X=1 --workload c2
run c2_1m X=1 --workload c2 --mesh 1m --particles 32000000 --steps 10
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); ph=j["roofline"].get("phases",{}); print("%-28s ms/step %8.4f value %.3e frac %.3f  ps %.3f rest %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"], ph.get("push_search",{}).get("ms") or 0, ph.get("rebuild_scatter",{}).get("ms") or 0))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
