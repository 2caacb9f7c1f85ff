#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/us_fwd_ab.txt
python -m pytest tests/test_unet_gpu.py tests/test_headline_parity_gpu.py tests/test_sampling_gpu.py -x -q -m gpu 2>&1 | tail -3
tr() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>$O/us_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $*', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/us_fwd_ab.txt
}
sa() {
  env "$@" python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/us_fwd_ab.txt
}
for rep in 1 2 3; do
tr VILLAN_US_FWD_PRESPLIT=0
tr VILLAN_US_FWD_PRESPLIT=1
sa VILLAN_US_FWD_PRESPLIT=0
sa VILLAN_US_FWD_PRESPLIT=1
done
cat $O/us_fwd_ab.txt; tail -2 $O/us_err.txt
