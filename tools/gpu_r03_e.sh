#!/bin/bash
# wgrad k32: parity, then per-shape and per-step A/B on the same box
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "weight_gradient or grouped" > $O/t_wk32.log 2>&1
grep -E "FAILED|passed|failed|Error" $O/t_wk32.log | tail -n 15
for rep in 1 2; do
  VD_WGRAD_K32_OFF=1 timeout 300 python tools/wgrad_bx3_bench.py 2>&1 | grep "^B=" | sed 's/^/old /'
  timeout 300 python tools/wgrad_bx3_bench.py 2>&1 | grep "^B=" | sed 's/^/k32 /'
done | sort -s -k2,5 | cut -c1-30,88-140
for rep in 1 2; do
  for cfg in "VD_WGRAD_K32_OFF=1" "VD_NOP=1"; do
    for sw in "" "--serial-wgrad"; do
      env $cfg timeout 300 python3 bench.py --mode train --no-cpu --no-exact --no-roofline $sw 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $sw', d['ms_per_step'])"
    done
  done
done
