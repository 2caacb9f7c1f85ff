#!/bin/bash
# k32 (16x16x32 MFMA) convolution: parity tests, then A/B per shape and per training step on the same box
O=gpurun_out/r03
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "split_precision_conv3x3 or upsample_conv or folded or tile_choice" > $O/t_k32.log 2>&1
tail -n 5 $O/t_k32.log
for rep in 1 2; do
  VD_BX3_K32_OFF=1 timeout 300 python tools/shape_probe.py conv3 > $O/probe_old_$rep.txt 2>&1
  timeout 300 python tools/shape_probe.py conv3 > $O/probe_k32_$rep.txt 2>&1
done
paste -d'\n' $O/probe_old_1.txt $O/probe_k32_1.txt $O/probe_old_2.txt $O/probe_k32_2.txt | grep -v "^$" | sort -s -k2,4 | head -60
for rep in 1 2; do
  VD_BX3_K32_OFF=1 timeout 300 python3 bench.py --mode train --no-cpu --no-exact --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['ms_per_step'])"
  timeout 300 python3 bench.py --mode train --no-cpu --no-exact --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k32', d['ms_per_step'])"
  VD_BX3_K32_KEEP_HUGE=1 timeout 300 python3 bench.py --mode train --no-cpu --no-exact --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k32+huge', d['ms_per_step'])"
done
