#!/bin/bash
# round 5: K-range cap / workgroup target of the grouped weight gradients, re-tuned for the pre-split kernel (same box, one process per setting)
O=gpurun_out/r05h
mkdir -p $O
for cap in 96 128 192 256; do
  for tgt in 384 448 512; do
    VD_WGRAD_GROUP_KCAP=$cap VD_WGRAD_K32_TARGET=$tgt timeout 200 python tools/wgrad_ps_ab.py 2>/dev/null | grep -E "grouped 3x3 wgrad.*(ps|k32) " | sed "s/^/cap=$cap target=$tgt  /" >> $O/knobs.txt
  done
done
cat $O/knobs.txt | grep " ps "
