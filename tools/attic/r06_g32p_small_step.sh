#!/bin/bash
# round 6: the small-tile 1x1 variant inside the training step and the sampler
O=gpurun_out/r06
mkdir -p $O
rm -f $O/g32p_small_step.txt
tr() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $*', d['value'], d['ms_per_step'])" >> $O/g32p_small_step.txt
}
sa() {
  env "$@" python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/g32p_small_step.txt
}
for rep in 1 2 3; do
tr VD_G32P_SMALL=0
tr VD_G32P_SMALL=2 VD_G32P_SMALL_WGS=2
sa VD_G32P_SMALL=0
sa VD_G32P_SMALL=2 VD_G32P_SMALL_WGS=2
done
cat $O/g32p_small_step.txt
