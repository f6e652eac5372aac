#!/bin/bash
# same-box A/B of two builds of the library on the 2-D literal: tools/r05_ab_2d.sh <old.so> <new.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PP_BENCH_NO_EXTRAS=1
run() { lib=$1; shift; PUMIPIC_HIP_LIB=$lib python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f  frac %.3f push_search %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['phases']['push_search']['ms']))"; }
for rep in 1 2 3; do
  echo "2dc3 old"; run $1 --workload 2dc3 --steps 40
  echo "2dc3 new"; run $2 --workload 2dc3 --steps 40
done
