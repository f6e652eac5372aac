#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_i
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py -x -q -m gpu -k "steps_3d or fused or comm" 2>&1 | tail -3
timeout 200 python tools/fuzz_search.py 60 3 2>&1 | tail -1
export PP_BENCH_NO_COLD=1
for wl in c3 c2; do for t in "" "--no-origin-trust"; do
timeout 300 python bench.py --workload $wl --no-cpu-baseline $t > $O/b.json 2>/dev/null
python - <<PY
import json
j=json.load(open("$O/b.json"))
print("$wl $t", "ms/step %.4f"%j["ms_per_step"], "frac %.3f"%j["roofline"]["frac"], j["roofline"].get("phases",{}).get("push_search",{}).get("ms"), j.get("origin_trust",{}).get("unmoved_without_test_last_step"))
PY
done; done
