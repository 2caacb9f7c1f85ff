#!/bin/bash
run() { timeout 900 python bench.py --mode train --steps 40 --warmup 10 --no-exact --no-cpu --no-roofline $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_submit_ms_per_step'))"; }
for i in 1 2; do
echo "== eager launches"; run ""
echo "== --graph-step 1"; run "--graph-step 1"
done
