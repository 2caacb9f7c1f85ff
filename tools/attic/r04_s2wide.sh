#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "stride2 or S2 or stride_2" 2>&1 | tail -4 > gpurun_out/s2wide.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "celebahq" 2>&1 | tail -3 >> gpurun_out/s2wide.txt
for r in 1 2; do for v in 0 1; do
  echo "VD_BX3_S2_OFF=$v" >> gpurun_out/s2wide.txt
  VD_BX3_S2_OFF=$v STEP_BENCH_TOP=30 timeout 300 python tools/step_bench.py celebahq256 2>&1 | grep -E "ms/step|S2|<128, 4|<64, 4" >> gpurun_out/s2wide.txt
done; done
cat gpurun_out/s2wide.txt
