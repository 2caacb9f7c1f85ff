#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/g32p_min.txt
for r in 1 2; do for m in 192 128 96 64; do
  echo "VD_G32P_MIN_TILES=$m" >> gpurun_out/g32p_min.txt
  VD_G32P_MIN_TILES=$m STEP_BENCH_TOP=4 python tools/step_bench.py ldm64 2>&1 | grep -E "ms/step|gemm_bx3_kernel|gemm1x1" >> gpurun_out/g32p_min.txt
done; done
cat gpurun_out/g32p_min.txt
