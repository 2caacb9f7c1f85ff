#!/bin/bash
# round 4: persistent 16x16x32 1x1 kernel -- parity, per-shape A/B, training / sampling A/B
mkdir -p gpurun_out/r04
out=gpurun_out/r04/g32p_ab.txt
: > $out
timeout 1500 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "1x1 or attention or activation_products" > gpurun_out/r04/t_g32p.log 2>&1
tail -n 3 gpurun_out/r04/t_g32p.log >> $out
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -x >> gpurun_out/r04/t_g32p.log 2>&1
tail -n 2 gpurun_out/r04/t_g32p.log >> $out
for rep in 1 2; do
for cfg in "VD_G32P_OFF=1" "VD_NOP=1"; do
  echo "== $cfg gemm" >> $out
  env $cfg python tools/shape_probe.py gemm 2>&1 | grep conv1x1 >> $out
done
done
for rep in 1 2 3; do
for cfg in "VD_G32P_OFF=1" "VD_NOP=1"; do
    r=$(env $cfg python bench.py --mode train --no-exact --no-cpu --no-roofline --steps 30 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$cfg train $r" >> $out
done
done
for cfg in "VD_G32P_OFF=1" "VD_NOP=1"; do
    r=$(env $cfg python bench.py --mode sample --no-cpu --no-roofline --no-secondary --sample-images 512 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['sample_ddpm1000_images_per_sec'])")
    echo "$cfg sample $r" >> $out
done
cat $out
