#!/bin/bash
# SQ counters of the intersection-mode walk (k_search_mt3): instructions per visited element
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_mt_sq
mkdir -p $O
cd $R; export TMPDIR=/tmp PP_BENCH_NO_COLD=1 PP_BENCH_PREWARM=0
A="bench.py --workload c2mt --steps 3 --warmup 2 --no-cpu-baseline --no-also"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -o p -- python3 $A > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq2 -o p -- python3 $A > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv -d $O/sq3 -o p -- python3 $A > $O/sq3.log 2>&1
python3 $R/tools/pmc_summary.py $O | grep -A30 -E "k_search_mt3" | head -80
tail -3 $O/sq3.log
rm -rf $O/sq1 $O/sq2 $O/sq3
