# kernel stats of the 2-D step with and without split records (lab build), one box, three rounds
R=$GRAFT_REPO_ROOT
export PP_BENCH_NO_EXTRAS=1 PUMIPIC_HIP_LIB=$R/pumi-pic_amd/libpumipic_hip_lab.so TOPN=3
for rep in 1 2 3; do
bash $R/tools/r06_kt.sh rec32 $R/bench.py --no-cpu-baseline --workload 2dc3 --steps 40
PP_REC_SPLIT=1 bash $R/tools/r06_kt.sh split2 $R/bench.py --no-cpu-baseline --workload 2dc3 --steps 40
done
