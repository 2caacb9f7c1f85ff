#!/bin/bash
run() { timeout 600 python bench.py --mode train --steps 40 --warmup 10 --no-exact --no-cpu --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
echo "== resident positions"; run
echo "== host positions, pinned non-blocking"; VD_BENCH_HOST_POSITIONS=1 run
echo "== host positions, pageable (round-2 behaviour)"; VD_BENCH_HOST_POSITIONS=1 VILLAN_PAGEABLE_H2D=1 run
done
