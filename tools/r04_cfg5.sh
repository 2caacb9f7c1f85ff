#!/bin/bash
# round 4: config #5 work (flash attention, split-K 1x1): parity tests + step breakdown
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/cfg5_t.txt
python -m pytest tests/test_unet_gpu.py tests/test_config5_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/cfg5_t.txt
STEP_BENCH_TOP=24 STEP_BENCH_SHAPES=24 python tools/step_bench.py ldm64 >> gpurun_out/cfg5_t.txt 2>&1
VD_GEMM_BX3_SPLIT_OFF=1 STEP_BENCH_TOP=3 python tools/step_bench.py ldm64 2>&1 | grep "ms/step" >> gpurun_out/cfg5_t.txt
cat gpurun_out/cfg5_t.txt
