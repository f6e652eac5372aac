timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in c3 2dc3; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])"; done
timeout 900 python bench.py --workload c3 --mesh 1m --particles 32000000 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 1m 32M', d['value'], d['ms_per_step'])"
cd pumi-pic_amd/drivers; timeout 900 ./ps_combo160 50000 10000000 2 0 -i 10 -s 64 2>&1 | grep -E "migrate " | head -1
