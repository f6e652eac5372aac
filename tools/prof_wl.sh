#!/bin/bash
# Per-kernel time + HBM traffic of one bench workload (run on the GPU box):
#   bash tools/prof_wl.sh <tag> <bench args...>      -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc.txt}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/kt.log 2>&1
if [ "${PMC:-1}" = 1 ]; then
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc/fetch -o p -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pmc/write -o p -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
fi
cd $R
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/kt
[ "${PMC:-1}" = 1 ] && python tools/pmc_summary.py $O/pmc > $O/pmc.txt 2>&1
find $O/pmc -name "*.csv" -delete 2>/dev/null
python - "$O" <<'PY'
import csv, sys, re
O = sys.argv[1]
rows = list(csv.DictReader(open(O + "/kernel_stats.csv")))
pm = {}
try:
    cur = None
    for line in open(O + "/pmc.txt"):
        if not line.startswith(" "):
            cur = line.strip(); pm[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", line)
            if m: pm[cur][m.group(1)] = float(m.group(2))
except FileNotFoundError:
    pass
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    n = r["Name"]; k = n[:60]
    c = pm.get(k, {})
    # FETCH_SIZE is in KiB and reports half the bytes on gfx950 (calibrated); WRITE_SIZE exact
    rd = c.get("FETCH_SIZE", 0) * 1024 / 0.5002 / 1e6
    wr = c.get("WRITE_SIZE", 0) * 1024 / 1e6
    print("%-58s calls %5s avg %8.1f us  %5.1f%%  rd %7.1f MB wr %7.1f MB" % (n[:58], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, rd, wr))
PY
