timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in c3 2dc3; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_c3 -o p -- python3 $R/bench.py --workload c3 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/gpurun_out/kt_c3 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_c3.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("  %-45s calls/step=%.1f us/step=%.1f" % (r["Name"][:45], int(r["Calls"])/23, float(r["TotalDurationNs"])/23e3))
PY
