#!/bin/bash
# the GPU suite under every remaining knob + the four fuzzers (round-3 final code) -> profiles/r03_matrix.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_matrix
mkdir -p $O
cd $R
( bash tools/gpu_test_matrix.sh ) > $O/matrix.txt 2>&1
for f in rebuild search migrate picpart; do
  ( timeout 200 python tools/fuzz_$f.py 120 ) > $O/fuzz_$f.txt 2>&1
  echo "fuzz_$f: $(tail -1 $O/fuzz_$f.txt)" >> $O/matrix.txt
done
cat $O/matrix.txt
