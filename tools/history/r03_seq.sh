#!/bin/bash
# one step's kernel sequence with start offsets: tools/r03_seq.sh <workload>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_seq; mkdir -p $O; wl=${1:-c5}
extra=""; [ $wl = c5 ] && extra="--mesh 1m --particles 32000000 --steps 6"
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1 PP_BENCH_NO_COLD=1
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_$wl -o p -- python3 $R/bench.py --workload $wl $extra --no-cpu-baseline --no-scale-ref > $O/kt_$wl.log 2>&1
t=$(find $O/kt_$wl -name "*kernel_trace.csv" | head -1); m=$(find $O/kt_$wl -name "*memory_copy_trace.csv" | head -1)
python3 - "$t" "$m" <<'PY'
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:70]) for r in csv.DictReader(open(sys.argv[1]))]
try:
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes","")) for r in csv.DictReader(open(sys.argv[2]))]
except Exception as e: print("no copy trace",e)
rows.sort()
idx=[i for i,r in enumerate(rows) if "k_push_walk_rows" in r[2]]
a,b=idx[-3],idx[-2]
t0=rows[a][0]
prev_end=rows[a][0]
for s,e,n in rows[a:b+1]:
    print("%9.1f  gap %7.1f  dur %8.1f  %s"%((s-t0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,n)); prev_end=max(prev_end,e)
PY
rm -rf $O/kt_$wl
