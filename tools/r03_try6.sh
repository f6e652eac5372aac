#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try6
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_comm.py -x -q -m gpu -k "config5 or rehearsal or watchdog or rccl_single" ) > $O/pytest_comm.txt 2>&1
tail -8 $O/pytest_comm.txt
( time timeout 2400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or tet_c3" ) > $O/pytest_full.txt 2>&1
tail -8 $O/pytest_full.txt
( time timeout 1500 python bench.py --no-cpu-baseline ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][0])
print("ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "no trust", j.get("ms_per_step_no_origin_trust"))
print("scale_ref", json.dumps(j.get("scale_ref"), indent=1))
PY
