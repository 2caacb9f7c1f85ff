#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_f16_mode_gpu.py -q -m gpu -x -s > gpurun_out/r04/t_f16.log 2>&1
grep -E "parity|passed|failed|Error|error|assert" gpurun_out/r04/t_f16.log | tail -n 20
timeout 900 python bench.py --no-exact > gpurun_out/r04/bench_f16.json 2> gpurun_out/r04/bench_f16.err
tail -c 1500 gpurun_out/r04/bench_f16.json; grep -E "f16|train:|sample:" gpurun_out/r04/bench_f16.err | head
