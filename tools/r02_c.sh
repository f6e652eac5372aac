#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_c
mkdir -p $O
cd $R
for deg in 0.5 0.1 0.02; do
PP_BENCH_NO_COLD=1 timeout 300 python bench.py --no-cpu-baseline --deg $deg --remainder spread > $O/bench_c3_deg$deg.json 2>/dev/null
python - <<PY
import json
j=json.load(open("$O/bench_c3_deg$deg.json"))
print("deg $deg spread", round(j["ms_per_step"],4), j["roofline"]["phases"]["rebuild_scatter"]["ms"], j["rebuilds"])
PY
done
cd /tmp; export TMPDIR=/tmp
PP_BENCH_NO_COLD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-cpu-baseline --deg 0.02 --remainder spread > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_deg0.02.csv; rm -rf $O/kt
head -25 $O/kernel_stats_deg0.02.csv | cut -c1-60,200-400 
