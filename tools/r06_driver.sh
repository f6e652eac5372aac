#!/bin/bash
# the drop-in driver (drivers/pseudoXGCm: user lambdas through ps::parallel_for, unfused search / rebuild /
# scatter) at config-3 size on the 2-D literal mesh: wall time per step, its RecordTime table and kernel stats
#   tools/r06_driver.sh <tag> [numPtcls] [iterations]      env passes through; DRIVER=<exe> profiles another build of the
#   loop (tests/_refdrivers/pseudoXGCm: the reference source unchanged)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_driver
mkdir -p $O
tag=${1:-base}; NP=${2:-10000000}; IT=${3:-20}
make -C $R/pumi-pic_amd/drivers -s || exit 1
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
c, e, cl = pp.synth.annulus_tri()
pp.synth.write_mesh_bin("/tmp/annulus100k.bin", 2, c, e, cl)
print("mesh", len(e), "triangles")
PY
cd /tmp; export TMPDIR=/tmp
D=${DRIVER:-$R/pumi-pic_amd/drivers/pseudoXGCm}
timeout 600 $D /tmp/annulus100k.bin $NP 12 $IT 0.5 0 > $O/run_$tag.out 2> $O/run_$tag.err
echo "rc $?"; grep -E "iterations of pseudopush|RESULT" $O/run_$tag.out $O/run_$tag.err
grep -A40 -i "timing\|Summary" $O/run_$tag.out | head -60
if [ -z "$NOPROF" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o p -- $D /tmp/annulus100k.bin $NP 12 $IT 0.5 0 > $O/kt_$tag.log 2>&1
f=$(find $O/kt_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$tag.csv
t=$(find $O/kt_$tag -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" "ellipticalPush::push" > $O/gaps_$tag.txt 2>&1; cat $O/gaps_$tag.txt
python3 $R/tools/kernel_seq.py "$t" "ellipticalPush::push" > $O/seq_$tag.txt 2>&1
rm -rf $O/kt_$tag
python3 - <<PY
import csv
tot = 0
rows = list(csv.DictReader(open("$O/kernel_stats_$tag.csv")))
for r in rows: tot += float(r["TotalDurationNs"])
print("total kernel time per iteration: %.1f us" % (tot / 1e3 / $IT))
for r in rows[:24]:
    print("%-80s calls %5s avg %9.1f us  per-iter %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/$IT))
PY
fi
