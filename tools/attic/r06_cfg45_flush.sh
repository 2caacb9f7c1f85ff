#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/cfg45_flush.txt
run() {
  c=$1; shift
  env "$@" timeout 300 python bench.py --config $c --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c $*', d['value'], d['ms_per_step'])" >> $O/cfg45_flush.txt
}
for rep in 1 2; do
for j in 24 48 64 96 1000; do run celebahq256 VILLAN_WGRAD_FLUSH_JOBS=$j; done
for j in 24 48 64 96 1000; do run ldm64 VILLAN_WGRAD_FLUSH_JOBS=$j; done
done
cat $O/cfg45_flush.txt
