#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try8
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
