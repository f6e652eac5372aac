#!/bin/bash
# host timeline of the drop-in driver: HIP API calls beside the kernels for one step (where does the GPU wait for the host?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_driver
mkdir -p $O
NP=${1:-10000000}; IT=${2:-12}
make -C $R/pumi-pic_amd/drivers -s || exit 1
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import pumipic_amd_loader
pp = pumipic_amd_loader.load()
c, e, cl = pp.synth.annulus_tri()
pp.synth.write_mesh_bin("/tmp/annulus100k.bin", 2, c, e, cl)
PY
cd /tmp; export TMPDIR=/tmp
D=$R/pumi-pic_amd/drivers/pseudoXGCm
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/ht -o p -- $D /tmp/annulus100k.bin $NP 12 $IT 0.5 0 > $O/ht.log 2>&1
k=$(find $O/ht -name "*kernel_trace.csv" | head -1); a=$(find $O/ht -name "*hip_api_trace.csv" | head -1)
python3 $R/tools/host_timeline.py "$k" "$a" "ellipticalPush::push" > $O/host_timeline.txt 2>&1
rm -rf $O/ht
tail -120 $O/host_timeline.txt
