#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/us_dgrad_ab.txt
python -m pytest tests/test_unet_gpu.py tests/test_headline_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>$O/us_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/us_dgrad_ab.txt
}
for rep in 1 2 3; do
run VILLAN_US_DGRAD_PRESPLIT=0
run VILLAN_US_DGRAD_PRESPLIT=1
done
cat $O/us_dgrad_ab.txt; tail -2 $O/us_err.txt
