#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_f16_mode_gpu.py -q -m gpu -x -s > gpurun_out/r04/t_f16.log 2>&1
grep -E "parity|passed|failed|Error|error|assert" gpurun_out/r04/t_f16.log | tail -n 20
