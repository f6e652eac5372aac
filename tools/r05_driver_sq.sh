#!/bin/bash
# SQ counters of the drop-in driver's kernels (one --pmc pass, kernel trace only): tools/r05_driver_sq.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_driver_sq
mkdir -p $O
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
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o p -- $D /tmp/annulus100k.bin 10000000 12 10 0.5 0 > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -o p -- $D /tmp/annulus100k.bin 10000000 12 10 0.5 0 > $O/sq2.log 2>&1
python3 $R/tools/pmc_summary.py $O | grep -A16 -E "k_move_unpack|k_move_pack_rm|updatePtclPositions" | head -80
rm -rf $O/sq1 $O/sq2
