#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/t_all.log 2>&1
grep -E "FAILED|passed|failed|Error" $O/t_all.log | tail -n 15
STEP_BENCH_TOP=0 timeout 600 python tools/step_bench.py cifar10 2>&1 | grep -E "config #1|cifar10 B" 
