#!/bin/bash
O=gpurun_out/r05u
mkdir -p $O
for i in 1 2; do
for gs in 0 1; do
timeout 600 python bench.py --mode train --graph-step $gs --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph-step $gs:', d['value'], d['ms_per_step'], 'host submit', d.get('host_submit_ms_per_step'))" >> $O/out.txt
done
done
cat $O/out.txt
