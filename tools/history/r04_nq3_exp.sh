#!/bin/bash
# (round 5: some of the PP_* switches this script sets were deleted together with the variants they selected -- the
# script is kept as the record of how that round's numbers were taken; tools/gpu_test_matrix.sh is the live matrix)
# EXPERIMENT: duration of the FIRST row-major pack of a c3 run with 48-B records (PP_DBG_NQ3=1; the push then reads
# them at the wrong stride: timing of that one kernel only) against the 64-B records as shipped.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04_nq3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1 PP_BENCH_NO_COLD=1
for m in 0 1 0 1; do
  if [ $m = 1 ]; then export PP_DBG_NQ3=1; else unset PP_DBG_NQ3; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$m -o p -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/kt_$m.log 2>&1
  t=$(find $O/kt_$m -name "*kernel_trace.csv" | head -1)
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$t")) if "k_move_pack" in r["Kernel_Name"]]
print("48-B records" if $m else "64-B records", " ".join("%s %.1f" % (r["Kernel_Name"][28:46], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows[:3]), "us")
PY
  rm -rf $O/kt_$m
done
