#!/bin/bash
# round 4: persistent 16x16x32 convolution -- parity, per-shape A/B (round-3 kernel / persistent / persistent + LDS-DMA weights), training step A/B
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "persistent or tile_choice or unsplit or upsample_conv or channel_sums or folded" > gpurun_out/r04/t_k32p.log 2>&1
tail -n 4 gpurun_out/r04/t_k32p.log; grep "parity\] persistent" gpurun_out/r04/t_k32p.log
for rep in 1 2; do
for cfg in "VD_K32P_OFF=1" "VD_NOP=1" "VD_K32P_DMA=1"; do
  echo "== $cfg conv3" >> gpurun_out/r04/k32p_ab.txt
  env $cfg python tools/shape_probe.py conv3 2>&1 | grep -E "@32|@16" >> gpurun_out/r04/k32p_ab.txt
done
done
for cfg in "VD_K32P_OFF=1" "VD_NOP=1" "VD_K32P_DMA=1"; do
  echo "== $cfg conv3w" >> gpurun_out/r04/k32p_ab.txt
  env $cfg python tools/shape_probe.py conv3w 2>&1 | grep conv3x3 >> gpurun_out/r04/k32p_ab.txt
done
for rep in 1 2; do
for cfg in "VD_K32P_OFF=1" "VD_NOP=1" "VD_K32P_DMA=1"; do
  for extra in "" "--serial-wgrad"; do
    r=$(env $cfg python bench.py --mode train --no-exact --no-cpu --no-roofline --steps 30 $extra 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$cfg $extra $r" >> gpurun_out/r04/k32p_ab.txt
  done
done
done
cat gpurun_out/r04/k32p_ab.txt
