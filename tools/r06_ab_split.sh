#!/bin/bash
# same-box A/B of the split 2-D records (lab build: PP_REC_SPLIT=1 turns them on; default = the 32-B record + side word)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PP_BENCH_NO_EXTRAS=1 PUMIPIC_HIP_LIB=$R/pumi-pic_amd/libpumipic_hip_lab.so
run() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f  frac %.3f push_search %.4f rebuild_scatter %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['phases']['push_search']['ms'], d['roofline']['phases']['rebuild_scatter']['ms']))"; }
for rep in 1 2 3; do
  echo "2dc3 32-B records + side word"; run --workload 2dc3 --steps 40
  echo "2dc3 split records (two arrays of 16-B halves)"; PP_REC_SPLIT=1 run --workload 2dc3 --steps 40
done
