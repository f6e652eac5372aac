#!/bin/bash
# SQ counters of the c3 step's kernels (k_push_walk_rowsq, k_move_pack_rm, ...): three --pmc passes, kernel trace only
# (the program directly after `--`): tools/r06_sq_c3.sh [workload]
W=${1:-c3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_sq_$W
mkdir -p $O
cd $R; export TMPDIR=/tmp PP_BENCH_NO_COLD=1 PP_BENCH_PREWARM=0
A="bench.py --workload $W --steps 5 --warmup 3 --no-cpu-baseline --no-also --no-scale-ref"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o p -- python3 $A > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq2 -o p -- python3 $A > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT --kernel-trace --output-format csv -d $O/sq3 -o p -- python3 $A > $O/sq3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM --kernel-trace --output-format csv -d $O/sq4 -o p -- python3 $A > $O/sq4.log 2>&1
python3 $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1
head -150 $O/summary.txt
tail -2 $O/sq4.log
rm -rf $O/sq1 $O/sq2 $O/sq3 $O/sq4
