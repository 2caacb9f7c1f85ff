#!/bin/bash
# round 6: stage-program variants of conv3_k32p_kernel<.., PS> on one box, interleaved: VD_K32P_LAG = 0 round 5's lockstep program, 1 lagging wave half,
# 2 spread vector-memory issue (FINE)
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2; do
  for v in 0 1; do
    VD_K32P_LAG=$v python tools/k32p_probe.py --check > $O/k32p_lag${v}_$rep.txt 2>&1
  done
done
VD_K32P_LAG=1 K32P_LIB=tools/diag/libvillan_hip_k32p_stamps.so python tools/k32p_probe.py --stamps > $O/k32p_lag1_stamps.txt 2>&1
for f in 4 8 16; do
VD_K32P_LAG=1 K32P_LIB=tools/diag/libvillan_hip_k32p_var.so VD_K32P_FLAGS=$f python tools/k32p_probe.py > $O/k32p_fl_lag1_$f.txt 2>&1
done
paste <(cut -c1-36 $O/k32p_lag0_1.txt) <(cut -c22-36 $O/k32p_lag1_1.txt) <(cut -c22-36 $O/k32p_lag0_2.txt) <(cut -c22-120 $O/k32p_lag1_2.txt)
grep "sum of" $O/k32p_lag?_?.txt $O/k32p_fl_lag1_*.txt
grep -A8 "128->128 @32 mode 2\|512->256 @16 mode 2" $O/k32p_lag1_stamps.txt
