#!/bin/bash
# the committed default line + the c3 kernel stats / gaps of the final code
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --no-cpu-baseline --workload c3 --steps 40 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3.csv
t=$(find $O/kt -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_c3.txt 2>&1
rm -rf $O/kt
cat $O/gaps_c3.txt | head -8
python3 -c "
import json
d=json.load(open('$O/bench_c3.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
"
