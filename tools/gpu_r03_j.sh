#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "grouped" 2>&1 | tail -2
for rep in 1 2; do
for t in "VD_WGRAD9=0" "VD_NOP=1" "VD_WGRAD9_KCAP=128" "VD_WGRAD9_SLAB_STEPS=0" "VD_WGRAD9_SLAB_STEPS=48" "VD_WGRAD9_TARGET=512"; do
  env $t timeout 300 python3 bench.py --mode train --no-cpu --no-exact --serial-wgrad 2>$O/w9.err | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
print('$t', l['ms_per_step'], [(k['kernel'][:24], k['ms']) for k in d['train_step_kernels'] if 'wgrad9' in k['kernel'] or 'wgrad_bx3_group_kernel<32, 0' in k['kernel'] or 'wgrad_bx3_group_kernel<16, 0' in k['kernel']])"
done
done
