#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/streams_ab.txt
for rep in 1 2; do
for n in 4 3 5 6 8; do
timeout 300 python bench.py --mode sample --sample-steps 150 --sample-images 1024 --sample-streams $n --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $n', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/streams_ab.txt
done
done
cat $O/streams_ab.txt
