#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_train_sample_gpu.py -q -m gpu -x -k "deterministic" > $O/t_det.log 2>&1
grep -E "Fatal|fault|FAILED|passed|failed|Error" $O/t_det.log | head -5
timeout 1200 python -m pytest tests/test_headline_parity_gpu.py tests/test_train_sample_gpu.py -q -m gpu -x > $O/t_two.log 2>&1
grep -E "Fatal|fault|FAILED|passed|failed|Error" $O/t_two.log | head -5
bash tools/gpu_r03_e.sh
