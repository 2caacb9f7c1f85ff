#!/bin/bash
# round 3, first GPU pass: new parity tests, the default bench line (size + parity object), --gpus 2 on the one GPU (gloo)
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "batch128" -s > $O/t_b128.log 2>&1
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "grouped" > $O/t_grouped.log 2>&1
VD_BENCH_DETAIL=$O/bench_detail.json timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
VD_BENCH_BACKEND=gloo VD_BENCH_DETAIL=$O/bench2_detail.json timeout 600 python3 bench.py --gpus 2 --steps 5 --warmup 2 --sample-steps 50 --no-secondary --no-exact > $O/bench_gpus2.json 2> $O/bench_gpus2.err
tail -3 $O/t_b128.log $O/t_grouped.log; wc -c $O/bench_default.json; tail -2 $O/bench_gpus2.json | cut -c1-600
