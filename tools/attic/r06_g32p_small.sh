#!/bin/bash
# round 6: 128 x 128 tiles (NI = 2) in the persistent 1x1 kernel: never / always, one or two workgroups per CU
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2; do
VD_G32P_SMALL=0 python tools/g32p_bm_ab.py 2>/dev/null | sed "s/^/small=0       /"
VD_G32P_SMALL=2 VD_G32P_SMALL_WGS=1 python tools/g32p_bm_ab.py 2>/dev/null | sed "s/^/small=2 wgs=1 /"
VD_G32P_SMALL=2 VD_G32P_SMALL_WGS=2 python tools/g32p_bm_ab.py 2>/dev/null | sed "s/^/small=2 wgs=2 /"
done > $O/g32p_small.txt
cut -c1-190 $O/g32p_small.txt
