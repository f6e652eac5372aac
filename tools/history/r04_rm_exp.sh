#!/bin/bash
# (round 5: some of the PP_* switches this script sets were deleted together with the variants they selected -- the
# script is kept as the record of how that round's numbers were taken; tools/gpu_test_matrix.sh is the live matrix)
# EXPERIMENT: duration of the FIRST k_move_pack of a c3 run with (1) row-major staging records + block-level
# transpose, (2) block-level transpose only, (0) as shipped.  Mode 1 corrupts the structure: timing only.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04_rm; mkdir -p $O
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1 PP_BENCH_NO_COLD=1
for m in 0 2 1 0 2 1; do
  PP_DBG_RM=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$m -o p -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 0 > $O/kt_$m.log 2>&1
  t=$(find $O/kt_$m -name "*kernel_trace.csv" | head -1)
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$t")) if "k_move_pack" in r["Kernel_Name"]]
print("mode $m:", " ".join("%.1f" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows[:4]), "us (first calls of the pack)")
PY
  rm -rf $O/kt_$m
done
