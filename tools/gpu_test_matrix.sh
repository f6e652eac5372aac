# GPU suite under every knob that selects between SUPPORTED paths: tile width, the two fused-kernel
# variants, the record-fed push off (second pass of the re-layout run at once); layout sort: the radix-pass
# scan as its own launches, the one-pass 11-bit sort off (at all / beyond 64 tiles), chunk widths as their
# own launch, no on-device pass skipping / pass prediction; rebuild: the histogram cleared by a fill, the
# slot -> element table written in every re-layout, gyroScatter behind the rebuild instead of riding in it,
# the totals by D2H copy / by event instead of the polled stamp, the tail after the host sync instead of
# speculatively, eager x_tgt zeros; atomic / flat scatter forms, SoA placement rules off, the reference's
# reshuffle decision off on both sides.  The whole suite passes in all configurations.
for cfg in "PP_TILE_P=4" "PP_TILE_P=16" "PP_TILE_P=32" "PP_WALK_QUEUE=1" "PP_WALK_QUEUE=0" "PP_NO_LAZY_UNPACK=1" \
           "PP_NO_FUSED_SORT=1" "PP_NO_WIDE_SORT=1" "PP_NO_WIDE_SORT_BIG=1" "PP_NO_FUSED_WIDTHS=1" \
           "PP_NO_RS_SKIP=1" "PP_NO_RS_PREDICT=1" \
           "PP_NO_PREZERO=1" "PP_EAGER_SLOT_ELEM=1" "PP_NO_SCATTER_RIDE=1" "PP_NO_DIRECT_TOTALS=1" \
           "PP_NO_POLL_TOTALS=1" "PP_NO_SPEC_REBUILD=1" "PP_NO_LAZY_ZERO=1" \
           "PP_SCATTER_ATOMIC=1" "PP_SCATTER_FLAT=1" "PP_NO_STRIDE_SPREAD=1" "PP_NO_MEMBER_SKEW=1" \
           "PP_TEST_SHUFFLING=0" "PP_MT_PACKED=0" "PP_MT_PER_LANE=3" "PP_MT_START_BATCH=1" \
           "PP_NO_REC_PAD=1" "PP_NO_RM_RECORDS=1" "PP_NO_PAIR_FETCH=1" "PP_RM_WIDE=0" "PP_NO_TABLES_SLOTS_MERGE=1" "PP_NO_COUNT_MERGE=1" "PP_NO_HOT_ROW=1"; do
  echo "== $cfg"; env $cfg timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
done
