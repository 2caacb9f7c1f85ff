#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
rm -f $O/flush_sweep.txt
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d.get('host_submit_ms_per_step'))" >> $O/flush_sweep.txt
}
for rep in 1 2; do
for j in 24 12 16 32 48; do
run VILLAN_WGRAD_FLUSH_JOBS=$j
done
done
cat $O/flush_sweep.txt
