#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by +-5 %):
#   tools/r04_ab_lib.sh <other.so> <rounds> [bench args...]    -> kernel stats of each, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
other=$1; rounds=$2; shift 2
cp $R/pumi-pic_amd/libpumipic_hip.so /tmp/pp_new.so
cp $R/$other /tmp/pp_old.so
for i in $(seq 1 $rounds); do
  cp /tmp/pp_old.so $R/pumi-pic_amd/libpumipic_hip.so; TOPN=${TOPN:-4} bash $R/tools/r04_kt.sh old$i "$@"
  cp /tmp/pp_new.so $R/pumi-pic_amd/libpumipic_hip.so; TOPN=${TOPN:-4} bash $R/tools/r04_kt.sh new$i "$@"
done
cp /tmp/pp_new.so $R/pumi-pic_amd/libpumipic_hip.so
