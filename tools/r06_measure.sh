#!/bin/bash
# Round-6 refresh (run on the GPU box): PMC traffic of the c3 / 2dc3 / c2 / c4 steps with the calibration run (the c2 and
# c4 files were round-4 / round-1 measurements), kernel stats + gaps of c3 / 2dc3 / c2 / c2mt / the drop-in driver.
# Output: gpurun_out/r06_m/ (copied to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_m
mkdir -p $O
cd $R
[ -x tools/_ubs ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ub_stream.hip -o tools/_ubs
cd /tmp; export TMPDIR=/tmp PP_BENCH_NO_EXTRAS=1
kt() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o p -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/kt_$name.log 2>&1
  f=$(find $O/kt_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$name.csv
  t=$(find $O/kt_$name -name "*kernel_trace.csv" | head -1); python3 $R/tools/gpu_gaps.py "$t" > $O/gaps_$name.txt 2>&1
  rm -rf $O/kt_$name
}
WLS=${WLS:-"c3 2dc3 c2 c4"}   # (WLS="2dc3" bash tools/r06_measure.sh: one workload only)
for wl in $WLS; do
  case $wl in c4) ;; *) kt $wl --workload $wl --steps 40;; esac
done
case " $WLS " in *" c2 "*) kt c2mt --workload c2mt --steps 6;; esac
export PP_BENCH_NO_COLD=1 PP_BENCH_PREWARM=0
passw() { wl=$1; name=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$wl/$name -o p -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc_${wl}_$name.log 2>&1
}
for wl in $WLS; do
passw $wl fetch FETCH_SIZE TCC_EA0_RDREQ_sum
passw $wl write WRITE_SIZE TCC_EA0_WRREQ_sum
done
timeout 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc/cal_fetch -o p -- $R/tools/_ubs > $O/pmc_cal_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pmc/cal_write -o p -- $R/tools/_ubs > $O/pmc_cal_write.log 2>&1
cd $R
for wl in $WLS; do
  case $wl in c3) n=10000000; b=194;; 2dc3) n=10000000; b=162;; c2) n=10000000; b=69;; c4) n=1000000; b=489;; esac
  python tools/traffic_step.py $O $wl $n 13 $O/traffic_$wl.json pmc_$wl $b > $O/traffic_$wl.txt 2>&1
  rm -rf $O/pmc_$wl
done
rm -rf $O/pmc
for w in $WLS; do head -14 $O/traffic_$w.txt; done
