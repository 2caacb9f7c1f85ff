#!/bin/bash
O=gpurun_out/r05q
mkdir -p $O
timeout 900 python -m pytest tests/test_presplit_gpu.py -q -k "groupnorm_backward or network or mixed" 2>&1 | tail -4 > $O/tests.log
VD_GN_PS4_OFF=1 timeout 200 python tools/gn_ps_probe.py 2>/dev/null | grep -v amdgpu > $O/gn_probe_ps4off.txt
timeout 200 python tools/gn_ps_probe.py 2>/dev/null | grep -v amdgpu > $O/gn_probe_ps4on.txt
for i in 1 2; do
  VD_GN_PS4_OFF=1 timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ps4 off', d['value'], d['ms_per_step'])" >> $O/step.txt
  timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ps4 on ', d['value'], d['ms_per_step'])" >> $O/step.txt
done
cat $O/tests.log; head -3 $O/gn_probe_ps4off.txt; head -3 $O/gn_probe_ps4on.txt; cat $O/step.txt
