#!/bin/bash
# round 6, item 1(a): where does a stage of conv3_k32p_kernel<.., PS = true> go?  release times, ablations, stamps (one box, one lease)
O=gpurun_out/r06
mkdir -p $O
python tools/k32p_probe.py --check > $O/k32p_release.txt 2>&1
for f in 0 32 2 4 6 8 16 18 20 22; do
  K32P_LIB=tools/diag/libvillan_hip_k32p_var.so VD_K32P_FLAGS=$f python tools/k32p_probe.py > $O/k32p_var_$f.txt 2>&1
done
K32P_LIB=tools/diag/libvillan_hip_k32p_var.so VD_K32P_FLAGS=32 python tools/k32p_probe.py --check > $O/k32p_var_32_check.txt 2>&1
K32P_LIB=tools/diag/libvillan_hip_k32p_stamps.so python tools/k32p_probe.py --stamps > $O/k32p_stamps.txt 2>&1
K32P_LIB=tools/diag/libvillan_hip_k32p_stamps.so VD_K32P_FLAGS=32 python tools/k32p_probe.py --stamps > $O/k32p_stamps_32.txt 2>&1
python tools/k32p_probe.py > $O/k32p_release_again.txt 2>&1
tail -n 3 $O/k32p_release.txt $O/k32p_var_*.txt
cat $O/k32p_stamps.txt
