#!/bin/bash
# round 6: pixel-fragment read lead of conv3_k32p_kernel<.., PS>: one tile (round 5, diagnostic library -DVD_K32P_XD1) against two, same box, interleaved
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2 3; do
  VD_K32P_LAG=0 K32P_LIB=tools/diag/libvillan_hip_k32p_xd1.so python tools/k32p_probe.py --check > $O/k32p_xd1_$rep.txt 2>&1
  VD_K32P_LAG=0 python tools/k32p_probe.py --check > $O/k32p_xd2_$rep.txt 2>&1
done
paste <(cut -c1-36 $O/k32p_xd1_1.txt) <(cut -c22-36 $O/k32p_xd2_1.txt) <(cut -c22-36 $O/k32p_xd1_2.txt) <(cut -c22-36 $O/k32p_xd2_2.txt)  <(cut -c22-36 $O/k32p_xd1_3.txt) <(cut -c22-120 $O/k32p_xd2_3.txt)
grep "sum of" $O/k32p_xd?_?.txt
