#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_try12
mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests/test_ppmio.py -x -q ) > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
