#!/bin/bash
# the pad of the pre-split weight gradient, finer: 76 KB + pad must exceed 80 KB (one per CU), and what is left of the CU's 160 KB decides who else fits
# (conv3_sm_kernel at 8x8: 74 KB, at 4x4: 66 KB)
O=gpurun_out/r06
mkdir -p $O
rm -f $O/pad_sweep2.txt
run() {
  env "$@" timeout 300 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/pad_sweep2.txt
}
for rep in 1 2 3 4 5 6; do
for p in 12288 4608; do
run VD_WGRAD_PS_LDS_PAD=$p
done
done
cat $O/pad_sweep2.txt
