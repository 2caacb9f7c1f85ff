#!/bin/bash
for rep in 1 2; do
  for cfg in "3 384" "4 512" "6 768" "8 1024"; do
    set -- $cfg
    timeout 900 python3 bench.py --mode sample --no-cpu --no-roofline --no-secondary --sample-images $2 --sample-streams $1 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 images $2:', l['sample_ddpm1000_images_per_sec'], l['sample_seconds'])"
  done
done
