#!/bin/bash
# (round 5: some of the PP_* switches this script sets were deleted together with the variants they selected -- the
# script is kept as the record of how that round's numbers were taken; tools/gpu_test_matrix.sh is the live matrix)
# c2mt: the packed Moeller-Trumbore walk under its two knobs, and the unpacked form (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export PP_BENCH_NO_COLD=1
run() { env "$@" python bench.py --workload c2mt --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s ms/step %8.3f  visits/particle %.1f  ns/visit %s' % ('$*', j['ms_per_step'], j['walk']['elements_visited_per_particle'], j['walk']['ns_per_visited_element']))"; }
for pl in 1 2 4 8 16; do run PP_MT_PER_LANE=$pl; done
for sb in 4 8 12 16 24 32; do run PP_MT_START_BATCH=$sb; done
run PP_MT_PACKED=0
