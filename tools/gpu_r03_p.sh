#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py tests/test_cabi.py tests/test_train_sample_gpu.py tests/test_headline_parity_gpu.py -q -m gpu -x > gpurun_out/r03/t_p.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error|^E " gpurun_out/r03/t_p.log | tail -n 8
