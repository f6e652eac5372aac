#!/bin/bash
# generic A/B on one box: tools/r03_ab.sh "<env assignment of B>" [workloads...]; A = defaults
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03_ab; mkdir -p $O; cd $R; : > $O/ab.txt
B="$1"; shift; WLS="${@:-c3 2dc3}"
for i in 1 2 3; do
for cfg in "X=1" "$B"; do
  for wl in $WLS; do
    extra=""; [ $wl = c5 ] && extra="--mesh 1m --particles 32000000 --steps 10"
    printf "%s %s " "$cfg" $wl >> $O/ab.txt
    env $cfg PP_BENCH_NO_EXTRAS=1 timeout 600 python bench.py --workload $wl $extra --no-cpu-baseline --no-scale-ref 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], round(j['roofline']['frac'],4))" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
