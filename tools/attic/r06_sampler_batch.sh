#!/bin/bash
# round 6: how much does the sampler gain from larger forwards?  (chunk size through --batch: 128 = the protocol, 256 / 512 = what fusing 2 / 4 chunks into one
# launch sequence would give)  same box, 150-step loop on 1024 images
O=gpurun_out/r06
mkdir -p $O
rm -f $O/sampler_batch.txt
for rep in 1 2; do
for b in 128 256 512; do
for s in 4 2; do
python bench.py --mode sample --batch $b --sample-streams $s --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b streams $s', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/sampler_batch.txt
done
done
done
cat $O/sampler_batch.txt
