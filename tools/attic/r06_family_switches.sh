#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/family_switches.txt
run() {
  env "$@" timeout 300 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/family_switches.txt
}
for rep in 1 2 3; do
run VD_NOP=1
run VD_W1X1_WIDE=0
run VD_WGRAD_K32=0
run VD_G32P_BM256=0
done
cat $O/family_switches.txt
