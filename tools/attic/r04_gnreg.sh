#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "groupnorm or split_k" 2>&1 | tail -4 > gpurun_out/gnreg.txt
python -m pytest tests/test_unet_gpu.py tests/test_config5_fullsize_gpu.py tests/test_vqmodel.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/gnreg.txt
STEP_BENCH_TOP=8 python tools/step_bench.py ldm64 2>&1 | grep -E "ms/step|groupnorm" >> gpurun_out/gnreg.txt
STEP_BENCH_TOP=8 python tools/step_bench.py celebahq256 2>&1 | grep -E "ms/step|groupnorm" >> gpurun_out/gnreg.txt
cat gpurun_out/gnreg.txt
