#!/bin/bash
# round 6: the 12 MFMAs of a pixel tile product by product (diagnostic library -DVD_K32P_ORD) against channel tile by channel tile (shipped), same box
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2 3; do
  python tools/k32p_probe.py --check > $O/k32p_ord0_$rep.txt 2>&1
  K32P_LIB=tools/diag/libvillan_hip_k32p_ord.so python tools/k32p_probe.py --check > $O/k32p_ord1_$rep.txt 2>&1
done
grep "sum of\|DIFFER" $O/k32p_ord?_?.txt
