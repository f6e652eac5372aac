# GPU suite with every FALLBACK path forced (round 5: the A/B knobs of the tuning rounds are deleted; what is left
# selects code that is live anyway -- pp_internal.hpp, PP_LAB_ENV).  Needs the laboratory build:
#   make -C pumi-pic_amd/csrc lab && bash tools/gpu_test_matrix.sh
# column-loop / queued walk kernel whatever the dimension, pass 2 of the re-layout run at once, gyroScatter behind the
# rebuild instead of riding in it, the rebuild's tail after the host sync instead of speculatively, the atomic scatter
# (a ring map without a transpose), the unpacked Moeller-Trumbore walk, one column per record fetch, the over-full
# row through the common blocks, the reference's reshuffle decision off on both sides, the 2-D loop on split records
# (round 6: built, measured, not the default).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PUMIPIC_HIP_LIB=$R/pumi-pic_amd/libpumipic_hip_lab.so
CFGS=${CFGS:-"PP_WALK_QUEUE=1 PP_WALK_QUEUE=0 PP_NO_LAZY_UNPACK=1 PP_NO_SCATTER_RIDE=1 PP_NO_SPEC_REBUILD=1 PP_SCATTER_ATOMIC=1 PP_MT_PACKED=0 PP_NO_HOT_ROW=1 PP_TEST_SHUFFLING=0 PP_REC_SPLIT=1"}
for cfg in $CFGS; do   # (CFGS="PP_MT_PACKED=0 ..." bash tools/gpu_test_matrix.sh: some of them)
  echo "== $cfg"; env $cfg timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -6
done
