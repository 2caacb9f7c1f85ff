#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py tests/test_train_sample_gpu.py -q -m gpu -x > gpurun_out/r03/t_q.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error|^E " gpurun_out/r03/t_q.log | tail -n 6
VD_BENCH_DETAIL=gpurun_out/r03/q_detail.json timeout 900 python bench.py --steps 30 --warmup 10 --no-exact --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('sample_ddpm1000_images_per_sec'))"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/q_detail.json'))
for tab in ('train_step_kernels','sampler_step_kernels'):
    for r in d[tab]:
        if 'k32' in r['kernel']: print(tab[:7], r['kernel'], r['launches'], round(r['avg_us'],1), round(r['tflops'],1))
PY
