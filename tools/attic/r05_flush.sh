#!/bin/bash
O=gpurun_out/r05o
mkdir -p $O
for rnd in 1 2; do
for fj in 6 12 24 48 0; do
  VILLAN_WGRAD_FLUSH_JOBS=$fj timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flush_jobs=$fj round $rnd', d['value'], d['ms_per_step'])" >> $O/flush.txt
done
done
cat $O/flush.txt
