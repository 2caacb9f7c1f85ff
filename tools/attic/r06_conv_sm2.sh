#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
timeout 600 python -m pytest tests/test_conv_sm_gpu.py -x -q 2>&1 | grep -B30 "short test summary" | head -60
rm -f $O/conv_sm_step_ab.txt
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $*', d['value'], d['ms_per_step'])" >> $O/conv_sm_step_ab.txt
}
runs() {
  env "$@" python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/conv_sm_step_ab.txt
}
for rep in 1 2 3; do
run VD_CONV_SM_OFF=1
run VD_CONV_SM_OFF=0
done
for rep in 1 2; do
runs VD_CONV_SM_OFF=1
runs VD_CONV_SM_OFF=0
done
cat $O/conv_sm_step_ab.txt
