#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "elastic or rebuild" > $O/test_rebuild.log 2>&1; tail -30 $O/test_rebuild.log
for rem in spread last; do
PP_BENCH_NO_COLD=1 PP_SPEC_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --remainder $rem > $O/bench_c3_$rem.json 2> $O/bench_c3_$rem.err
grep "rebuild in place" $O/bench_c3_$rem.err | head -30
python - <<PY
import json
j=json.load(open("$O/bench_c3_$rem.json"))
print("$rem", round(j["ms_per_step"],4), j["roofline"]["phases"], j["rebuilds"])
PY
done
