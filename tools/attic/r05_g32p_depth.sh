#!/bin/bash
# same-box A/B of the 1x1 kernel's load lead (2 | 3 stages) + the 1x1 parity tests + a short bench per setting
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for rep in 1 2; do
  for dp in 2 3; do VD_G32P_DEPTH=$dp timeout 300 python tools/g32p_bm_ab.py 2>&1 | tail -12; done
done
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "1x1 or persistent or projection" 2>&1 | tail -5
for rep in 1 2; do
  for dp in 2 3; do echo "depth $dp"; VD_G32P_DEPTH=$dp timeout 600 python bench.py --steps 30 --warmup 5 --no-ddp-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; done
done
} > gpurun_out/r05_g32p_depth.log 2>&1
tail -60 gpurun_out/r05_g32p_depth.log
