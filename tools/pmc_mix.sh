#!/bin/bash
# Dynamic instruction mix of the bench kernels (run on the GPU box): bash tools/pmc_mix.sh <tag> <bench args>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
O=$R/gpurun_out/mix_$tag
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift; counters="$1"; shift
  timeout 240 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $O/pmc/$name -o p -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_$name.log 2>&1
}
pass a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "$@"
pass b "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "$@"
pass c "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "$@"
pass d "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" "$@"
cd $R
python tools/pmc_summary.py $O/pmc > $O/mix.txt 2>&1
rm -rf $O/pmc
grep -A40 "rowsq\|k_walk_pending" $O/mix.txt | head -120
