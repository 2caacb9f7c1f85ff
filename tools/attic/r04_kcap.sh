#!/bin/bash
# round 4: K-range cap of the grouped 3x3 weight gradient on config #4 (wide images: many slabs per layer at the default 128 steps)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/kcap.txt
for r in 1 2; do for k in 128 256 512 1024; do
  echo "VD_WGRAD_GROUP_KCAP=$k" >> gpurun_out/kcap.txt
  VD_WGRAD_GROUP_KCAP=$k STEP_BENCH_TOP=40 timeout 300 python tools/step_bench.py celebahq256 2>&1 | grep -E "ms/step|wgrad_bx3_group_kernel<32, 0, true>|wgrad_bx3_group_kernel<32, 2, true>" >> gpurun_out/kcap.txt
done; done
cat gpurun_out/kcap.txt
