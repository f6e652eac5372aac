#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; echo "all rc=$?" >> $O/test_all.log
tail -30 $O/test_all.log
timeout 200 python tools/fuzz_rebuild.py 60 1 > $O/fuzz_rebuild.log 2>&1; tail -3 $O/fuzz_rebuild.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err; cat $O/bench_c3.json
