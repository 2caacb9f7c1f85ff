#!/bin/bash
mkdir -p gpurun_out/r03
timeout 2700 python -m pytest tests -q -m gpu -x > gpurun_out/r03/t_all.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error" gpurun_out/r03/t_all.log | tail -n 6
