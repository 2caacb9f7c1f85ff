#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "groupnorm or flash or attention" 2>&1 | tail -4 > gpurun_out/cfg45b.txt
python -m pytest tests/test_unet_gpu.py tests/test_config5_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/cfg45b.txt
STEP_BENCH_TOP=16 python tools/step_bench.py celebahq256 >> gpurun_out/cfg45b.txt 2>&1
STEP_BENCH_TOP=8 python tools/step_bench.py ldm64 >> gpurun_out/cfg45b.txt 2>&1
cat gpurun_out/cfg45b.txt
