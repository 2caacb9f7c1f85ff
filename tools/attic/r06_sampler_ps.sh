#!/bin/bash
# round 6: the sampler's no-grad forward with GroupNorm folded into the convolutions' loaders (default) against pre-split producers + convolutions, same box
O=gpurun_out/r06
mkdir -p $O
rm -f $O/sampler_ps_ab.txt
for rep in 1 2; do
for v in 0 1; do
VILLAN_SAMPLER_PRESPLIT=$v python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>$O/sampler_ps_err_$v.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sampler_presplit=$v', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/sampler_ps_ab.txt
done
done
cat $O/sampler_ps_ab.txt; tail -3 $O/sampler_ps_err_1.txt
