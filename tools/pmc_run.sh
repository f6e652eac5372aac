#!/bin/bash
# Collect rocprofv3 PMC counters for a bench.py command in separate passes (SQ: 8 slots, TCC: 4;
# FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Every pass is
# bounded by `timeout`: a TA/TCP counter pass once hung a box for 25 minutes (r01).
# usage: tools/pmc_run.sh <outdir> <bench args...>      (run on the GPU box)
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="$*"
pass() {
  name=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- \
    python3 "$R/bench.py" --no-cpu-baseline $BENCH_ARGS > "$OUT/$name.log" 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
