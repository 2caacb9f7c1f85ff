#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py -q -m gpu -x -k "weight_gradient or wgrad or backward or group" > gpurun_out/r03/t_q.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error|^E " gpurun_out/r03/t_q.log | tail -n 6
for i in 1 2; do
VD_BENCH_DETAIL=gpurun_out/r03/q_detail.json timeout 900 python bench.py --mode train --steps 30 --warmup 10 --no-exact --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/q_detail.json'))
print(' | '.join(f"{r['kernel'][:34]} {r['ms']:.3f}" for r in d['train_step_kernels'] if 'wgrad_bx3_group' in r['kernel']))
PY
done
