#!/bin/bash
# ablation flags on the LAG / lockstep stage programs + static priority (diagnostic build, one box)
O=gpurun_out/r06
mkdir -p $O
for v in 0 1; do
  for f in 0 2 4 8 16 64 128; do
    VD_K32P_LAG=$v K32P_LIB=tools/diag/libvillan_hip_k32p_var.so VD_K32P_FLAGS=$f python tools/k32p_probe.py > $O/k32p_fl_lag${v}_$f.txt 2>&1
  done
done
grep "sum of" $O/k32p_fl_lag*.txt
