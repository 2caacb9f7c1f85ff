#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py tests/test_cabi.py -q -m gpu -x -k "groupnorm or unet or oracle or fused or cabi or abi" > gpurun_out/r03/t_l.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error|parity\] fused" gpurun_out/r03/t_l.log | tail -n 8
for f in 1 0 1 0; do
  echo "== VILLAN_FUSE_GN_BWD=$f"
  VILLAN_FUSE_GN_BWD=$f timeout 600 python bench.py --steps 30 --warmup 10 --mode train --no-exact --no-cpu --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
