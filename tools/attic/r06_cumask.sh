#!/bin/bash
# round 6 (review item 5b): a CU mask for the weight-gradient side stream, same box, interleaved
O=gpurun_out/r06
mkdir -p $O
rm -f $O/cumask_ab.txt
run() {
  env "$@" python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>$O/cumask_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/cumask_ab.txt
}
for rep in 1 2; do
run VD_NOP=1
run VILLAN_WGRAD_CU_MASK=77777777
run VILLAN_WGRAD_CU_MASK=55555555
run VILLAN_WGRAD_CU_MASK=0f0f0f0f
run VILLAN_WGRAD_CU_MASK=7f7f7f7f
done
cat $O/cumask_ab.txt; tail -3 $O/cumask_err.txt
