#!/bin/bash
for rep in 1 2 3; do
  for g in 0 1; do
    timeout 300 python3 bench.py --mode train --no-cpu --no-exact --no-roofline --graph-step $g 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph-step $g', l['ms_per_step'], l['host_submit_ms_per_step'], l['final_loss'])"
  done
done
