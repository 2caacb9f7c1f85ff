#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/sampler_streams.txt
: > $out
for st in 4 6 8 4 6 8; do
    r=$(python bench.py --mode sample --no-cpu --no-roofline --no-secondary --sample-images 1536 --sample-streams $st 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])")
    echo "streams $st: $r" >> $out
done
cat $out
