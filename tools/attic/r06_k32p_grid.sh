#!/bin/bash
# round 6: the persistent convolution on fewer workgroups than CUs (power-bound: does it lose anything?  do the CUs it leaves help the other streams?)
O=gpurun_out/r06
mkdir -p $O
rm -f $O/k32p_grid.txt
tr() {
  env "$@" timeout 300 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $*', d['value'], d['ms_per_step'])" >> $O/k32p_grid.txt
}
sa() {
  env "$@" timeout 300 python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/k32p_grid.txt
}
for rep in 1 2; do
for g in 256 248 240 224 192; do
tr VD_K32P_GRID=$g
sa VD_K32P_GRID=$g
done
done
cat $O/k32p_grid.txt
