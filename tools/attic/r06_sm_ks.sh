#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/sm_imgs2_ab.txt
tr() {
  env "$@" timeout 300 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $*', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/sm_imgs2_ab.txt
}
sa() {
  env "$@" timeout 300 python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/sm_imgs2_ab.txt
}
for rep in 1 2 3; do
tr VD_CONV_SM_IMGS2=0
tr VD_CONV_SM_IMGS2=1
sa VD_CONV_SM_IMGS2=0
sa VD_CONV_SM_IMGS2=1
done
cat $O/sm_imgs2_ab.txt
