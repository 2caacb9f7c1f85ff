#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/prio_ab.txt
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>$O/prio_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/prio_ab.txt
}
for rep in 1 2 3; do
run VD_NOP=1
run VD_BENCH_MAIN_PRIO=1
done
cat $O/prio_ab.txt; tail -2 $O/prio_err.txt
