#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_f
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; echo "all rc=$?" >> $O/test_all.log
tail -8 $O/test_all.log
timeout 200 python tools/fuzz_rebuild.py 90 1 > $O/fuzz_rebuild.log 2>&1; tail -3 $O/fuzz_rebuild.log
timeout 200 python tools/fuzz_search.py 40 1 > $O/fuzz_search.log 2>&1; tail -2 $O/fuzz_search.log
timeout 200 python tools/fuzz_migrate.py 40 1 > $O/fuzz_migrate.log 2>&1; tail -2 $O/fuzz_migrate.log
PP_BENCH_NO_COLD=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
j=json.load(open("$O/bench_c3.json"))
print("c3", round(j["ms_per_step"],4), j["roofline"]["phases"], j["rebuilds"])
PY
