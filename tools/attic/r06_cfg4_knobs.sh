#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/cfg4_knobs.txt
run() {
  env "$@" timeout 300 python bench.py --config celebahq256 --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/cfg4_knobs.txt
}
for rep in 1 2; do
run VD_NOP=1
run VD_WGRAD_GROUP_KCAP=64
run VD_WGRAD_GROUP_KCAP=256
run VD_WGRAD_GROUP_KCAP=512
run VD_WGRAD_GROUP_TARGET=512
run VD_WGRAD_GROUP_TARGET=1536
run VILLAN_WGRAD_FLUSH_JOBS=8
run VILLAN_WGRAD_FLUSH_JOBS=48
done
cat $O/cfg4_knobs.txt
