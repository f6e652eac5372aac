#!/bin/bash
# A/B of the rebuild's launch chain against the one-kernel layout (PP_COOP_LAYOUT=1) and the side-queue
# scatter (PP_SIDE_SCATTER=1): barrier micro-benchmark, the GPU suite, bench lines and kernel traces of
# c3 and of the c4 stress point.  Output: gpurun_out/r02_coop/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_coop
mkdir -p $O
cd $R
[ -x tools/_ubb ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_gridbar.hip -o tools/_ubb
timeout 120 tools/_ubb > $O/ub_gridbar.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest.txt
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>$O/bench_$name.err | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print('%-28s ms/step %8.4f value %.3e frac %.3f' % ('$name', j['ms_per_step'], j['value'], j['roofline']['frac']))
"; }
( b c3
PP_COOP_LAYOUT=1 b c3_coop
PP_SIDE_SCATTER=1 b c3_side
b c4 --workload c4
PP_COOP_LAYOUT=1 b c4_coop --workload c4
b c4_csr --workload c4 --structure csr
b c5_1m --workload c5 --mesh 1m --particles 32000000 --steps 10
b 2dc3 --workload 2dc3 ) > $O/ab.txt 2>&1
cd /tmp; export TMPDIR=/tmp
kt() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/kt_$name.log 2>&1
  f=$(find $O/kt_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv
  t=$(find $O/kt_$name -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_$name.txt 2>&1
  rm -rf $O/kt_$name
}
kt c3
kt c4_1M --workload c4 --steps 30
kt c4_1M_csr --workload c4 --structure csr --steps 30
cd $R
cat $O/ub_gridbar.txt $O/pytest.txt $O/ab.txt
