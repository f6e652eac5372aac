#!/bin/bash
# round-2 first GPU pass: new communicator tests, the whole GPU suite, baseline bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/test_comm.log 2>&1; echo "comm rc=$?" >> $O/test_comm.log
tail -25 $O/test_comm.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_comm.py > $O/test_all.log 2>&1; echo "all rc=$?" >> $O/test_all.log
tail -5 $O/test_all.log
timeout 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err; cat $O/bench_c3.json
timeout 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5_1rank.json 2> $O/bench_c5.err; tail -2 $O/bench_c5.err; cat $O/bench_c5_1rank.json
timeout 300 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null; cat $O/bench_c2.json
python bench.py --gpus 2 --steps 5; echo "gpus2 rc=$?"
