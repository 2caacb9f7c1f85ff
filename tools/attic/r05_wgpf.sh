#!/bin/bash
O=gpurun_out/r05s
mkdir -p $O
timeout 900 python -m pytest tests/test_presplit_gpu.py -q -k "weight_gradient or network or mixed" 2>&1 | tail -4 > $O/tests.log
timeout 300 python tools/wgrad_ps_ab.py 2>/dev/null | grep -E "grouped|bit-identical" > $O/ab.txt
for i in 1 2 3; do
timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/tests.log $O/ab.txt
