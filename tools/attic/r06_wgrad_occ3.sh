#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/wgrad_occ3_ab.txt
python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>$O/wgrad_occ_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/wgrad_occ3_ab.txt
}
for rep in 1 2 3; do
run VD_WGRAD_K32_LDS_PAD=0
run VD_WGRAD_K32_LDS_PAD=36864
run VD_WGRAD_PS_LDS_PAD=0
done
cat $O/wgrad_occ3_ab.txt
