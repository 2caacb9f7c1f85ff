#!/bin/bash
# same-box A/B of two builds of vd_presplit.hip (without / with the cross-step fragment prefetch of the pre-split weight gradient)
O=gpurun_out/r05t
mkdir -p $O
cp villandiffusion_amd/csrc/vd_presplit.hip /tmp/vd_presplit_new.hip
for rnd in 1 2; do
for v in nocarry carry; do
  if [ $v = nocarry ]; then cp tools/ab/vd_presplit_nocarry.hip villandiffusion_amd/csrc/vd_presplit.hip; else cp /tmp/vd_presplit_new.hip villandiffusion_amd/csrc/vd_presplit.hip; fi
  make -C villandiffusion_amd/csrc > $O/make_$v.log 2>&1
  timeout 300 python tools/wgrad_ps_ab.py 2>/dev/null | grep -E "grouped.* ps " | sed "s/^/$v $rnd: /" >> $O/ab.txt
  timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $rnd: step', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
done
cp /tmp/vd_presplit_new.hip villandiffusion_amd/csrc/vd_presplit.hip
cat $O/ab.txt
