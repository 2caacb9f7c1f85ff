#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py tests/test_cabi.py -q -m gpu -x > gpurun_out/r03/t_o.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error" gpurun_out/r03/t_o.log | tail -n 8
run() { timeout 600 python bench.py --steps 30 --warmup 10 --no-exact --no-cpu --no-roofline --no-secondary --sample-images 128 --sample-streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('sample_ddpm1000_images_per_sec'))"; }
for i in 1 2; do
echo "== stride-2 bf16x3"; run
echo "== VD_BX3_S2_OFF=1"; VD_BX3_S2_OFF=1 run
done
