#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try13
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_all.txt
PP_WALK_QUEUE=1 timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
PP_WALK_QUEUE=0 timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
