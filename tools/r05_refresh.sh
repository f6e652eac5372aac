#!/bin/bash
# Round-5 refresh after the 32-B records / XCD-aware pack: kernel stats + gaps of c3, the default bench line, the drop-in
# driver, the eight virtual ranks of config 5; then tools/r05_measure.sh (PMC traffic of c3 / 2dc3, stats of 2dc3 / c2mt).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_refresh
mkdir -p $O
cd $R
PP_BENCH_NO_EXTRAS=1 TOPN=12 bash tools/r05_kt.sh c3 $R/bench.py --no-cpu-baseline --workload c3 --steps 40 > $O/kt_c3.txt 2>&1
cp gpurun_out/r05_kt/kernel_stats_c3.csv $O/
timeout 900 python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 600 $O/bench_c3.json
NOPROF= bash tools/r05_driver.sh final 10000000 100 > $O/driver.txt 2>&1
cp gpurun_out/r05_driver/kernel_stats_final.csv gpurun_out/r05_driver/seq_final.txt gpurun_out/r05_driver/gaps_final.txt $O/ 2>/dev/null
timeout 900 python3 bench.py --workload c5 --virtual-ranks 8 --particles 32000000 --no-cpu-baseline > $O/c5_virtual8.json 2> $O/c5_virtual8.err
tail -c 400 $O/c5_virtual8.json
TOPN=14 bash tools/r05_kt.sh c5v8 $R/bench.py --workload c5 --virtual-ranks 8 --particles 32000000 --no-cpu-baseline --steps 4 --warmup 3 > $O/kt_c5v8.txt 2>&1
cp gpurun_out/r05_kt/kernel_stats_c5v8.csv $O/
bash tools/r05_measure.sh > $O/measure.txt 2>&1
tail -30 $O/measure.txt
