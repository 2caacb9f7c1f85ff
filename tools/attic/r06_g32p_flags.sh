#!/bin/bash
# round 6: where a tile of the persistent 1x1 kernel spends its time: timing-only ablations (diagnostic build, VD_G32P_FLAGS; WRONG results)
O=gpurun_out/r06
mkdir -p $O
for f in 0 2 4 8 16 32 36 6 38 62; do
G32P_LIB=tools/diag/libvillan_hip_g32p_var.so VD_G32P_FLAGS=$f python tools/g32p_bm_ab.py 2>&1 | sed "s/^/flags=$f /" > $O/g32p_flags_$f.txt
done
python tools/g32p_bm_ab.py > $O/g32p_flags_release.txt 2>&1
cat $O/g32p_flags_release.txt $O/g32p_flags_*.txt | cut -c1-175
