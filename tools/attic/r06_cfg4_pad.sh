#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/cfg4_pad_ab.txt
run() {
  env "$@" python bench.py --config celebahq256 --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 8 --warmup 3 2>$O/cfg4_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/cfg4_pad_ab.txt
}
for rep in 1 2; do
run VD_WGRAD_BX3_LDS_PAD=0
run VD_WGRAD_BX3_LDS_PAD=12288
run VD_WGRAD_BX3_LDS_PAD=40960
done
cat $O/cfg4_pad_ab.txt; tail -2 $O/cfg4_err.txt
