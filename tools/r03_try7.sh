#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try7
mkdir -p $O
cd $R
NCCL_DEBUG=INFO timeout 600 python -m pytest tests/test_gpu_comm.py -x -q -m gpu -k "rccl_single" > $O/pytest_rccl.txt 2>&1
grep -n "WARN\|error\|Error\|passed\|failed" $O/pytest_rccl.txt | head -30
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
