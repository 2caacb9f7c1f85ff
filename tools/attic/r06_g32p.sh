#!/bin/bash
# round 6: 1x1 kernel with 16-byte activation loads (shipped) against round 5's dword loads (tools/diag/libvillan_hip_g32p_r05.so), same box, interleaved
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2; do
  G32P_LIB=tools/diag/libvillan_hip_g32p_r05.so python tools/g32p_bm_ab.py > $O/g32p_r05_$rep.txt 2>&1
  python tools/g32p_bm_ab.py > $O/g32p_r06_$rep.txt 2>&1
done
cat $O/g32p_r05_1.txt $O/g32p_r06_1.txt; tail -1 $O/g32p_r05_2.txt $O/g32p_r06_2.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "1x1 or gemm or attn or attention or conv1x1 or unet" 2>&1 | tail -4
