#!/bin/bash
# Round-1 final measurements (run on the GPU box): bench lines, per-kernel rocprof stats, HBM
# traffic counters with their calibration run, micro-benchmarks.  Output: gpurun_out/r01_e/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r01_e
mkdir -p $O
cd $R
[ -x tools/_ubg ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_gather.hip -o tools/_ubg
[ -x tools/_ubs ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_stream.hip -o tools/_ubs
[ -x tools/_ubc ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_copy.hip -o tools/_ubc
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --remainder spread --no-cpu-baseline > $O/bench_c2_spread.json 2>/dev/null
timeout 300 python bench.py --workload 2d --no-cpu-baseline > $O/bench_2d.json 2>/dev/null
timeout 300 python bench.py --workload 2dc3 --no-cpu-baseline > $O/bench_2dc3.json 2>/dev/null
timeout 300 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
timeout 900 python bench.py --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/bench_c2_1mtet_32M.json 2>/dev/null
timeout 600 python bench.py --workload c4 > $O/bench_c4_1Me_1Mp_scs.json 2>/dev/null
timeout 600 python bench.py --workload c4 --structure csr --no-cpu-baseline > $O/bench_c4_1Me_1Mp_csr.json 2>/dev/null
timeout 600 python bench.py --workload c4 --c4-elems 50000 --particles 50000000 --steps 10 --no-cpu-baseline > $O/bench_c4_50ke_50Mp_scs.json 2>/dev/null
timeout 600 python bench.py --workload c4 --structure csr --c4-elems 50000 --particles 50000000 --steps 10 --no-cpu-baseline > $O/bench_c4_50ke_50Mp_csr.json 2>/dev/null
timeout 120 tools/_ubg > $O/ub_gather.txt 2>&1
( echo "# 10 M slots (partly MALL-resident)"; timeout 120 tools/_ubc 10485760; echo "# 100 M slots (HBM)"; timeout 200 tools/_ubc 100000000 ) > $O/ub_copy.txt 2>&1
timeout 120 tools/_ubs > $O/ub_stream.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -o p -- python3 $R/bench.py --no-cpu-baseline > $O/kt_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2d -o p -- python3 $R/bench.py --workload 2d --no-cpu-baseline > $O/kt_2d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2dc3 -o p -- python3 $R/bench.py --workload 2dc3 --no-cpu-baseline > $O/kt_2dc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o p -- python3 $R/bench.py --workload c3 --no-cpu-baseline > $O/kt_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o p -- python3 $R/bench.py --workload c4 --c4-elems 50000 --particles 50000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_c4.log 2>&1
pass() { name=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$name -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc_$name.log 2>&1
}
pass fetch FETCH_SIZE TCC_EA0_RDREQ_sum
pass write WRITE_SIZE TCC_EA0_WRREQ_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
passw() { wl=$1; name=$2; shift 2
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$wl/$name -o p -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc_${wl}_$name.log 2>&1
}
for wl in c3 2d c4; do
passw $wl fetch FETCH_SIZE TCC_EA0_RDREQ_sum
passw $wl write WRITE_SIZE TCC_EA0_WRREQ_sum
done
# calibration: the streaming skeleton moves exactly 37 B read + 32 B written per slot
timeout 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc/cal_fetch -o p -- $R/tools/_ubs > $O/pmc_cal_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pmc/cal_write -o p -- $R/tools/_ubs > $O/pmc_cal_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
for k in c2 2d 2dc3 c3 c4; do f=$(find $O/kt_$k -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$k.csv; done
rm -rf $O/kt_c2 $O/kt_2d $O/kt_2dc3 $O/kt_c3 $O/kt_c4
find $O/pmc -name "*.csv" ! -name "*counter_collection.csv" -delete
python tools/traffic_json.py $O c2 10000000 $O/traffic_c2.json > /dev/null
python tools/traffic_json.py $O c3 10000000 $O/traffic_c3.json pmc_c3 > /dev/null
python tools/traffic_json.py $O 2d 10000000 $O/traffic_2d.json pmc_2d > /dev/null
python tools/traffic_json.py $O c4 1000000 $O/traffic_c4.json pmc_c4 > /dev/null
find $O/pmc_c3 $O/pmc_2d $O/pmc_c4 -name "*.csv" -delete
cat $O/bench_c2.json; grep -B1 -A6 "rowsq\|s_rows<8, 4>\|pending" $O/pmc_summary.txt | head -80
