#!/bin/bash
# kernel stats of one workload on the box: tools/r03_kt.sh <workload> [pattern]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_kt; mkdir -p $O; wl=${1:-c3}
extra=""; [ $wl = c5 ] && extra="--mesh 1m --particles 32000000 --steps 10"
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$wl -o p -- python3 $R/bench.py --workload $wl $extra --no-cpu-baseline --no-scale-ref --steps 20 > $O/kt_$wl.log 2>&1
f=$(find $O/kt_$wl -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$wl.csv
t=$(find $O/kt_$wl -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" $( [ $wl = 2dc3 ] && echo k_push_walk_rows ) > $O/gaps_$wl.txt 2>&1
rm -rf $O/kt_$wl
python3 - $O/kernel_stats_$wl.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-64s calls %5s avg %9.1f us"%(r['Name'][:64],r['Calls'],float(r['AverageNs'])/1e3))
PY
cat $O/gaps_$wl.txt | head -4
tail -2 $O/kt_$wl.log | cut -c1-300
