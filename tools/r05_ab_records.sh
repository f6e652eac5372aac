#!/bin/bash
# same-box A/B of two builds of the library (64-B records of the pseudoXGCm type against 32-B records + side word):
# c3, 2dc3 and the c5 share (rank-of-8 population), alternating runs.   tools/r05_ab_records.sh <old.so> <new.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_ab
mkdir -p $O
cd $R
export PP_BENCH_NO_EXTRAS=1
run() { lib=$1; shift; PUMIPIC_HIP_LIB=$lib python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for rep in 1 2 3; do
  for wl in c3 2dc3; do
    echo "$wl old"; run $1 --workload $wl --steps 40
    echo "$wl new"; run $2 --workload $wl --steps 40
  done
done
for rep in 1 2; do
  echo "c5 (one rank) old"; run $1 --workload c5 --particles 32000000 --steps 8
  echo "c5 new"; run $2 --workload c5 --particles 32000000 --steps 8
done
