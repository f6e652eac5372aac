#!/bin/bash
# (round 5: some of the PP_* switches this script sets were deleted together with the variants they selected -- the
# script is kept as the record of how that round's numbers were taken; tools/gpu_test_matrix.sh is the live matrix)
# the knobs that meet the over-full row's blocks (pp_ps::hot), on the parity file + the 10 M-particle c3 property test
for cfg in "PP_NO_HOT_ROW=1" "PP_TILE_P=4" "PP_TILE_P=16" "PP_TILE_P=32" "PP_RM_WIDE=0" "PP_NO_COUNT_MERGE=1" "PP_NO_SPEC_REBUILD=1" \
           "PP_NO_POLL_TOTALS=1" "PP_NO_DIRECT_TOTALS=1" "PP_NO_FUSED_WIDTHS=1" "PP_NO_WIDE_SORT=1" "PP_NO_PAIR_FETCH=1"; do
  echo "== $cfg"; env $cfg timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not config1 and not c4 and not intersection" 2>&1 | grep -E "passed|failed|error" | tail -2
done
