#!/bin/bash
# round 4: stride-2 weight gradient on the split-precision kernel incl. 32-pixel segments of wide outputs: parity + config #4 / #2 A/B (VILLAN_WGRAD_S2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "stride2" 2>&1 | tail -5 > gpurun_out/s2wgrad.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "celebahq or ldm" 2>&1 | tail -3 >> gpurun_out/s2wgrad.txt
for r in 1 2; do for v in none wide all; do
  echo "VILLAN_WGRAD_S2=$v" >> gpurun_out/s2wgrad.txt
  VILLAN_WGRAD_S2=$v STEP_BENCH_TOP=40 timeout 300 python tools/step_bench.py celebahq256 2>&1 | grep -E "ms/step|S2|, 4, " >> gpurun_out/s2wgrad.txt
done; done
for v in wide all wide all; do
  echo "cifar10 VILLAN_WGRAD_S2=$v" >> gpurun_out/s2wgrad.txt
  VILLAN_WGRAD_S2=$v python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/s2wgrad.txt
done
cat gpurun_out/s2wgrad.txt
