#!/bin/bash
# does the clock pre-warm help or hurt the c3 line?  ms_per_step for several pre-warm lengths, and a 100-step run
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for pw in 0 0.1 0.3 1.0; do
  PP_BENCH_PREWARM=$pw python $R/bench.py --no-also --no-scale-ref --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('prewarm $pw: ms_per_step %.4f  cold pass %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['cold_clocks']['ms_per_step'], d['roofline']['kernel_ms']))"
done
done
PP_BENCH_PREWARM=0.3 python $R/bench.py --no-also --no-scale-ref --no-cpu-baseline --steps 100 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
t=d['cold_clocks']['host_ms_of_each_step_then_closing_barrier']
print('100 steps: ms_per_step %.4f cold %.4f; cold trace every 10th:' % (d['ms_per_step'], d['cold_clocks']['ms_per_step']), t[1::10])"
