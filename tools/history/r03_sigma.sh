cd ${GRAFT_REPO_ROOT:-.}
for sg in 0 262144 65536 16384 4096 1024; do
  extra=""; [ $sg != 0 ] && extra="--sigma $sg"
  printf "sigma %s: " $sg
  PP_BENCH_NO_EXTRAS=1 timeout 300 python bench.py --workload c3 $extra --no-cpu-baseline --no-scale-ref 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], round(j['roofline']['frac'],4), j['config']['workload'][:120])"
done
