#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_fid_gpu.py -q -m gpu -s -k "split_precision_conv3x3 or upsample_conv or folded or tile_choice or fid or general_conv or pool3 or bilinear or inception" > $O/t_k32.log 2>&1
grep -E "parity.*(Inception|FID|conv [0-9])|FAILED|passed|failed|Error" $O/t_k32.log | tail -n 40
for rep in 1 2; do
  VD_BX3_K32_OFF=1 timeout 300 python tools/shape_probe.py conv3 2>&1 | grep -E "@(32|16)\^2" | grep -v "   3" > $O/probe_old_$rep.txt
  VD_BX3_K32_FLAGS=1 timeout 300 python tools/shape_probe.py conv3 2>&1 | grep -E "@(32|16)\^2" | grep -v "   3" > $O/probe_nostag_$rep.txt
  timeout 300 python tools/shape_probe.py conv3 2>&1 | grep -E "@(32|16)\^2" | grep -v "   3" > $O/probe_k32_$rep.txt
done
paste -d'\n' $O/probe_old_1.txt $O/probe_nostag_1.txt $O/probe_k32_1.txt $O/probe_old_2.txt $O/probe_nostag_2.txt $O/probe_k32_2.txt | grep -v "^$" | sort -s -k2,4
for rep in 1 2; do
  for cfg in "VD_BX3_K32_OFF=1" "VD_BX3_K32_FLAGS=1" "VD_NOP=1"; do
    for sw in "" "--serial-wgrad"; do
      env $cfg timeout 300 python3 bench.py --mode train --no-cpu --no-exact --no-roofline $sw 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $sw', d['ms_per_step'])"
    done
  done
done
