#!/bin/bash
# round 4: one-launch chunked GroupNorm (chunks of a group meet inside the launch) -- parity + same-box A/B on config #4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "groupnorm" 2>&1 | tail -4 > gpurun_out/gnchunk1.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "celebahq or ldm or folded" 2>&1 | tail -3 >> gpurun_out/gnchunk1.txt
for r in 1 2; do for v in 0 1; do
  echo "VD_GN_CHUNK1_OFF=$v" >> gpurun_out/gnchunk1.txt
  VD_GN_CHUNK1_OFF=$v STEP_BENCH_TOP=8 timeout 300 python tools/step_bench.py celebahq256 2>&1 | grep -E "ms/step|groupnorm" >> gpurun_out/gnchunk1.txt
done; done
cat gpurun_out/gnchunk1.txt
