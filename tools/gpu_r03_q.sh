#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py tests/test_train_sample_gpu.py -q -m gpu -x > gpurun_out/r03/t_q.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error|^E " gpurun_out/r03/t_q.log | tail -n 6
run() { timeout 900 python bench.py --steps 30 --warmup 10 --no-exact --no-cpu --no-roofline --no-secondary --sample-images 128 --sample-streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('sample_ddpm1000_images_per_sec'))"; }
for i in 1 2; do
echo "== float4 split-K epilogue"; run
echo "== VD_SPLITK_EPI_SCALAR=1"; VD_SPLITK_EPI_SCALAR=1 run
done
