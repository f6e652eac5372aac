# (round 5: some of the PP_* switches this script sets were deleted together with the variants they selected -- the
# script is kept as the record of how that round's numbers were taken; tools/gpu_test_matrix.sh is the live matrix)
for cfg in "X=1" "PP_NO_WIDE_SORT_BIG=1" "PP_NO_WIDE_SORT=1" "PP_NO_FUSED_SORT=1" "PP_NO_RS_PREDICT=1" "PP_NO_SPEC_REBUILD=1" "PP_TILE_P=4" "PP_NO_SCATTER_RIDE=1" "PP_NO_POLL_TOTALS=1" "PP_TEST_SHUFFLING=0"; do
  echo "== $cfg"; env $cfg timeout 900 python -m pytest tests -m gpu -x -q -k "rebuild or fullsize or config5 or combo or migrate or lazy" 2>&1 | grep -E "passed|failed|error" | tail -2
done
for f in rebuild migrate; do timeout 150 python tools/fuzz_$f.py 90 2>&1 | tail -1; done
