#!/bin/bash
# round 4: persistent 16x16x32 convolution variants -- per-shape A/B and training step A/B (same box, interleaved)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/k32p_ab2.txt
: > $out
for rep in 1 2; do
for cfg in "VD_K32P_OFF=1" "VD_K32P_DMA=1" "VD_K32P_PIPE=1" "VD_K32P_DMA=1 VD_K32P_PIPE=1"; do
  echo "== $cfg conv3" >> $out
  env $cfg python tools/shape_probe.py conv3 2>&1 | grep -E "@32|@16" | grep -v "   3" >> $out
done
done
for cfg in "VD_K32P_OFF=1" "VD_K32P_DMA=1" "VD_K32P_DMA=1 VD_K32P_PIPE=1"; do
  echo "== $cfg conv3w" >> $out
  env $cfg python tools/shape_probe.py conv3w 2>&1 | grep conv3x3 >> $out
done
for rep in 1 2; do
for cfg in "VD_K32P_OFF=1" "VD_K32P_DMA=1" "VD_K32P_PIPE=1" "VD_K32P_DMA=1 VD_K32P_PIPE=1"; do
    r=$(env $cfg python bench.py --mode train --no-exact --no-cpu --no-roofline --steps 30 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$cfg train $r" >> $out
done
done
for cfg in "VD_K32P_OFF=1" "VD_K32P_DMA=1 VD_K32P_PIPE=1"; do
    r=$(env $cfg python bench.py --mode sample --no-cpu --no-roofline --no-secondary --sample-images 512 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['sample_ddpm1000_images_per_sec'])")
    echo "$cfg sample $r" >> $out
done
cat $out
