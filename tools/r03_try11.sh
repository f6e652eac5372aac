#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try11
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/test_gpu_picpart.py tests/test_golden.py tests/test_gpu_comm.py -x -q -m gpu ) > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
( timeout 150 python tools/fuzz_picpart.py 90 ) > $O/fuzz_picpart.txt 2>&1; tail -3 $O/fuzz_picpart.txt
