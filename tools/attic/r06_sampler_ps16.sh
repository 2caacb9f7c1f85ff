#!/bin/bash
# round 6: the sampler's no-grad forward with pre-split producers + pre-split convolutions at chosen image sizes only (VILLAN_NOGRAD_PRESPLIT=16 | 32 | 16,32)
O=gpurun_out/r06
mkdir -p $O
rm -f $O/sampler_ps16_ab.txt
for rep in 1 2; do
for v in "" 16 32 16,32; do
VILLAN_NOGRAD_PRESPLIT=$v python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>$O/sampler_ps16_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nograd_presplit=$v', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/sampler_ps16_ab.txt
done
done
cat $O/sampler_ps16_ab.txt; tail -3 $O/sampler_ps16_err.txt
