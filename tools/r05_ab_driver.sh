#!/bin/bash
# same-box A/B of the drop-in driver with two builds of the library: tools/r05_ab_driver.sh <libA.so> <libB.so> [iters]
R=${GRAFT_REPO_ROOT:-$(pwd)}
IT=${3:-100}
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
c, e, cl = pp.synth.annulus_tri()
pp.synth.write_mesh_bin("/tmp/annulus100k.bin", 2, c, e, cl)
PY
D=$R/pumi-pic_amd/drivers
for i in 1 2 3; do for lib in $1 $2; do
  mkdir -p /tmp/ab_$lib; cp $R/pumi-pic_amd/$lib /tmp/ab_$lib/libpumipic_hip.so
  LD_LIBRARY_PATH=/tmp/ab_$lib $D/pseudoXGCm /tmp/annulus100k.bin 10000000 12 $IT 0.5 0 2>&1 >/dev/null | grep "iterations of pseudopush" | sed "s/^/$lib /"
done; done
