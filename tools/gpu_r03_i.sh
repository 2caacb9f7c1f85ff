#!/bin/bash
O=gpurun_out/r03p
mkdir -p $O
VD_BENCH_DETAIL=$O/bench_detail.json timeout 1200 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 900 python -m pytest tests/test_ddp_gpu.py -q -m gpu 2>&1 | tail -3
