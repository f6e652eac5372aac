# the record / lazy / parity / full-size tests under every forced fallback of the laboratory build (the whole suite
# under each: tools/gpu_test_matrix.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PUMIPIC_HIP_LIB=$R/pumi-pic_amd/libpumipic_hip_lab.so
for cfg in "PP_WALK_QUEUE=1" "PP_WALK_QUEUE=0" "PP_NO_LAZY_UNPACK=1" "PP_NO_SCATTER_RIDE=1" "PP_NO_SPEC_REBUILD=1" \
           "PP_NO_HOT_ROW=1" "PP_TEST_SHUFFLING=0"; do
  echo "== $cfg"; env $cfg timeout 900 python -m pytest tests/test_gpu_lazy.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
done
