#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
for rep in 1 2 3; do
for t in "VD_NOP=1" "VD_WGRAD_QUANT=1" "VD_WGRAD_QUANT=1 VD_WGRAD_SLAB_STEPS=0" "VD_WGRAD_QUANT=1 VD_WGRAD_SLAB_STEPS=24"; do
  for sw in "" "--serial-wgrad"; do
  env $t timeout 300 python3 bench.py --mode train --no-cpu --no-exact $sw 2>$O/q.err | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
print('$t $sw', l['ms_per_step'], [(k['kernel'][:28], k['ms']) for k in d['train_step_kernels'] if 'wgrad' in k['kernel'] and 'group' in k['kernel']][:4])"
  done
done
done
