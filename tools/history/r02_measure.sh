#!/bin/bash
# Round-2 measurements (run on the GPU box): bench lines, per-kernel rocprof stats, HBM traffic counters
# with their calibration run.  Output: gpurun_out/r02_m/ (copied to profiles/r02_*).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02_m
mkdir -p $O
cd $R
[ -x tools/_ubs ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_stream.hip -o tools/_ubs
# ---- bench lines (the first is the driver's command)
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --remainder spread --no-cpu-baseline > $O/bench_c3_spread.json 2>/dev/null
timeout 300 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null
timeout 300 python bench.py --workload 2d --no-cpu-baseline > $O/bench_2d.json 2>/dev/null
timeout 300 python bench.py --workload 2dc3 --no-cpu-baseline > $O/bench_2dc3.json 2>/dev/null
timeout 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5_1rank_100k_10M.json 2>/dev/null
timeout 900 python bench.py --workload c5 --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/bench_c5_1rank_1m_32M.json 2>/dev/null
timeout 900 python bench.py --workload c2 --mesh 1m --particles 32000000 --no-cpu-baseline --steps 10 > $O/bench_c2_1m_32M.json 2>/dev/null
timeout 600 python bench.py --workload c4 > $O/bench_c4_1Me_1Mp_scs.json 2>/dev/null
timeout 600 python bench.py --workload c4 --structure csr --no-cpu-baseline > $O/bench_c4_1Me_1Mp_csr.json 2>/dev/null
timeout 600 python bench.py --workload c4 --c4-elems 50000 --particles 50000000 --steps 10 --no-cpu-baseline > $O/bench_c4_50ke_50Mp_scs.json 2>/dev/null
timeout 600 python bench.py --workload c4 --structure csr --c4-elems 50000 --particles 50000000 --steps 10 --no-cpu-baseline > $O/bench_c4_50ke_50Mp_csr.json 2>/dev/null
for d in 2 3; do for st in scs csr; do
timeout 900 python bench.py --workload c4 --structure $st --c4-dist $d --c4-elems 50000 --particles 50000000 --steps 10 --no-cpu-baseline > $O/bench_c4_50ke_50Mp_${st}_dist$d.json 2>/dev/null
timeout 600 python bench.py --workload c4 --structure $st --c4-dist $d --no-cpu-baseline > $O/bench_c4_1Me_1Mp_${st}_dist$d.json 2>/dev/null
done; done
python bench.py --gpus 2 --steps 5 > $O/bench_gpus2_on_1gpu_box.txt 2>&1; echo "exit code $?" >> $O/bench_gpus2_on_1gpu_box.txt
# ---- kernel statistics of the same commands
cd /tmp; export TMPDIR=/tmp
kt() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/kt_$name.log 2>&1
  f=$(find $O/kt_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv
  t=$(find $O/kt_$name -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_$name.txt 2>&1
  rm -rf $O/kt_$name
}
kt c3
kt c2 --workload c2
kt 2dc3 --workload 2dc3
kt c5_1m_32M --workload c5 --mesh 1m --particles 32000000 --steps 10
kt c4 --workload c4 --c4-elems 50000 --particles 50000000 --steps 5 --warmup 2
kt c4_1M --workload c4 --steps 30
kt c4_1M_csr --workload c4 --structure csr --steps 30
# ---- HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes, whole step for c3, the roofline kernels for c2
export PP_BENCH_NO_COLD=1 PP_BENCH_PREWARM=0
passw() { wl=$1; name=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$wl/$name -o p -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc_${wl}_$name.log 2>&1
}
for wl in c3 c2; do
passw $wl fetch FETCH_SIZE TCC_EA0_RDREQ_sum
passw $wl write WRITE_SIZE TCC_EA0_WRREQ_sum
done
timeout 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc/cal_fetch -o p -- $R/tools/_ubs > $O/pmc_cal_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pmc/cal_write -o p -- $R/tools/_ubs > $O/pmc_cal_write.log 2>&1
cd $R
python tools/traffic_step.py $O c3 10000000 13 $O/traffic_c3.json pmc_c3 194 > $O/traffic_c3.txt 2>&1
python tools/traffic_json.py $O c2 10000000 $O/traffic_c2.json pmc_c2 > /dev/null 2>$O/traffic_c2.err
python tools/pmc_summary.py $O/pmc_c3 > $O/pmc_summary_c3.txt 2>&1
find $O -name "*.csv" ! -name "kernel_stats_*" -delete; find $O -type d -empty -delete
cat $O/bench_c3.json; cat $O/traffic_c3.txt | head -30
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); print("%-34s ms/step %8.4f value %.3e frac %.3f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
