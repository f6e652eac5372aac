#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try10
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_all.txt
timeout 600 python bench.py --no-cpu-baseline --no-scale-ref > $O/bench_c3.json 2> $O/bench_c3.err
timeout 600 python bench.py --no-cpu-baseline --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); print("%-20s ms/step %8.4f frac %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["roofline"]["frac"]))
    except Exception as e: print(os.path.basename(f), "FAILED", e); os.system("tail -5 %s" % f.replace(".json",".err"))
PY
( timeout 200 python tools/fuzz_rebuild.py 90 ) > $O/fuzz_rebuild.txt 2>&1; tail -3 $O/fuzz_rebuild.txt
( timeout 200 python tools/fuzz_search.py 90 ) > $O/fuzz_search.txt 2>&1; tail -3 $O/fuzz_search.txt
( timeout 200 python tools/fuzz_migrate.py 90 ) > $O/fuzz_migrate.txt 2>&1; tail -3 $O/fuzz_migrate.txt
for cfg in "PP_WALK_QUEUE=0" "PP_NO_LAZY_UNPACK=1" "PP_NO_SPEC_REBUILD=1" "PP_TEST_SHUFFLING=0" "PP_TILE_P=16"; do
  echo "== $cfg"; env $cfg timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
done
