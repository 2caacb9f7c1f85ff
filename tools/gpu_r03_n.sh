#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py -q -m gpu -k "grouped or batch128_backward" 2>&1 | tail -3
for rep in 1 2 3; do
for t in "VD_W1X1_WIDE=0" "VD_NOP=1" "VD_W1X1_WIDE_KCAP=32" "VD_W1X1_WIDE_KCAP=128"; do
  for sw in "" "--serial-wgrad"; do
  env $t timeout 300 python3 bench.py --mode train --no-cpu --no-exact $sw 2>$O/q.err | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
print('$t $sw', l['ms_per_step'], [(k['kernel'][:28], k['ms']) for k in d['train_step_kernels'] if 'wgrad1x1' in k['kernel']])"
  done
done
done
