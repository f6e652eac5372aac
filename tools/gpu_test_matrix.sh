# GPU suite under every kernel-selection knob (tile width, fused-kernel variant, streams policy,
# pending regions, direct vs staged move; round 2: rebuild decision off / shuffling off on both sides,
# eager x_tgt zeros, flat / atomic scatter forms, one-kernel layout, scatter on the side queue).  The whole suite passes in all configurations.
for cfg in "PP_TILE_P=4" "PP_TILE_P=16" "PP_TILE_P=32" "PP_WALK_QUEUE=1" "PP_WALK_QUEUE=0" "PP_NT=0" "PP_PEND_REGIONS=1" "PP_DIRECT_MOVE=1" "PP_SCATTER_ATOMIC=1" "PP_NO_STRIDE_SPREAD=1" "PP_NO_MEMBER_SKEW=1" "PP_NO_RS_SKIP=1" "PP_FUSE_PENDING=1" "PP_NO_SPEC_REBUILD=1" "PP_NO_RS_PREDICT=1" "PP_FLAT_WALK=1" "PP_WALK_OCC=3" "PP_TEST_SHUFFLING=0" "PP_NO_LAZY_ZERO=1" "PP_SCATTER_FLAT=1" "PP_COOP_LAYOUT=1" "PP_SIDE_SCATTER=1"; do
  echo "== $cfg"; env $cfg timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
done
