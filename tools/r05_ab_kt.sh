#!/bin/bash
# kernel stats of one workload under two builds of the library: tools/r05_ab_kt.sh <old.so> <new.so> <workload>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_ab
mkdir -p $O
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1
wl=${3:-c3}
for tag in old new; do
  lib=$1; [ $tag = new ] && lib=$2
  export PUMIPIC_HIP_LIB=$lib
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o p -- python3 $R/bench.py --no-cpu-baseline --workload $wl --steps 40 > $O/kt_${wl}_$tag.log 2>&1
  f=$(find $O/kt_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_${wl}_$tag.csv; rm -rf $O/kt_$tag
  echo "== $wl $tag"
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_${wl}_$tag.csv")))
for r in rows[:12]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
