#!/bin/bash
# The ORACLE (test infrastructure) under AddressSanitizer + UBSan, on the CPU: its known-answer tests, golden vectors,
# property tests and the file-format tests run against oracle/_build_asan/libppo.so (PPO_LIB).  ~2 minutes.
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/oracle -s asan || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
cd $R
env -u LD_PRELOAD LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 PPO_LIB=$R/oracle/_build_asan/libppo.so \
    OMP_NUM_THREADS=2 python -m pytest tests/test_golden.py tests/test_oracle_properties.py tests/test_oracle_kats.py \
    tests/test_picpart_oracle.py tests/test_ppmio.py tests/test_ptlio.py -q -m "not gpu"
