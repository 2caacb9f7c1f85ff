#!/bin/bash
# round 4: 16x16x32 weight gradient with two register sets (loads two K-steps ahead) vs the default kernel, same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "weight_gradient or wgrad or opt_in" 2>&1 | tail -3 > gpurun_out/wgk32.txt
for r in 1 2; do
for v in "VD_WGRAD_K32=0" "VD_WGRAD_K32=1" "VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512"; do
  echo "$v" >> gpurun_out/wgk32.txt
  env $v python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/wgk32.txt
done; done
VD_WGRAD_K32=1 STEP_BENCH_MICRO=0 STEP_BENCH_TOP=10 python tools/step_bench.py cifar10 2>&1 | grep -E "wgrad|ms/step" >> gpurun_out/wgk32.txt
VD_WGRAD_K32=0 STEP_BENCH_MICRO=0 STEP_BENCH_TOP=10 python tools/step_bench.py cifar10 2>&1 | grep -E "wgrad|ms/step" >> gpurun_out/wgk32.txt
cat gpurun_out/wgk32.txt
