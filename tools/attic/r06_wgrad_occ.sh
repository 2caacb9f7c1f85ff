#!/bin/bash
# round 6: grouped pre-split weight gradients at ONE workgroup per CU (dynamic LDS pad) against two: does the main stream flow better beside them?
O=gpurun_out/r06
mkdir -p $O
rm -f $O/wgrad_occ_ab.txt
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>$O/wgrad_occ_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/wgrad_occ_ab.txt
}
for rep in 1 2; do
run VD_WGRAD_PS_LDS_PAD=0
run VD_WGRAD_PS_LDS_PAD=12288
run VD_WGRAD_PS_LDS_PAD=12288 VD_WGRAD_K32_TARGET=256
run VD_WGRAD_PS_LDS_PAD=12288 VD_WGRAD_K32_TARGET=224
done
cat $O/wgrad_occ_ab.txt; tail -2 $O/wgrad_occ_err.txt
