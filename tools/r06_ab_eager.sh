#!/bin/bash
# same-box A/B of the pack's eager member loads (lab build: PP_PACK_EAGER=0 asks for a slot's members after its rank,
# =1 together with it; default: together when four slots in five are live)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PP_BENCH_NO_EXTRAS=1 PUMIPIC_HIP_LIB=$R/pumi-pic_amd/libpumipic_hip_lab.so
run() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f  frac %.3f push_search %.4f rebuild_scatter %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['phases']['push_search']['ms'], d['roofline']['phases']['rebuild_scatter']['ms']))"; }
for rep in 1 2 3; do
  for w in c3 2dc3; do
    echo "$w members after the rank"; PP_PACK_EAGER=0 run --workload $w --steps 40
    echo "$w members with the rank"; PP_PACK_EAGER=1 run --workload $w --steps 40
  done
done
