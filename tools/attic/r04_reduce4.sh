#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wgrad or weight_gradient or grouped" 2>&1 | tail -3 > gpurun_out/reduce4.txt
STEP_BENCH_TOP=12 timeout 300 python tools/step_bench.py celebahq256 2>&1 | grep -E "ms/step|wgrad" >> gpurun_out/reduce4.txt
for r in 1 2 3; do python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cifar10', d['value'], d['ms_per_step'])" >> gpurun_out/reduce4.txt; done
STEP_BENCH_TOP=6 timeout 300 python tools/step_bench.py ldm64 2>&1 | grep -E "ms/step" >> gpurun_out/reduce4.txt
cat gpurun_out/reduce4.txt
