#!/bin/bash
# A/B of the one-pass 11-bit layout sort + widths in the layout kernel + pre-cleared histogram (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_wide; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2; do
for cfg in "X=1" "PP_NO_POLL_TOTALS=1"; do
  for wl in c3 2dc3; do
    echo "== $cfg $wl" >> $O/ab.txt
    env $cfg PP_BENCH_NO_EXTRAS=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-scale-ref 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['frac'])" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1
for wl in c3 2dc3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$wl -o p -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-scale-ref --steps 40 > $O/kt_$wl.log 2>&1
f=$(find $O/kt_$wl -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$wl.csv
t=$(find $O/kt_$wl -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" $( [ $wl = 2dc3 ] && echo k_push_walk_rows ) > $O/gaps_$wl.txt 2>&1
python3 - "$t" > $O/seq_$wl.txt <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])),key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"][:60] for r in rows]
idx=[i for i,n in enumerate(names) if "k_push_walk_rows" in n]
a,b=idx[-3],idx[-2]
for r in rows[a:b]: print("%8.1f us  %s"%((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:90]))
PY
rm -rf $O/kt_$wl
done
head -20 $O/kernel_stats_c3.csv | cut -c1-150
cat $O/seq_c3.txt $O/seq_2dc3.txt
