#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
for a in "4 0 1" "16 0 1" "16 1 1"; do
  echo "== $a"; timeout 300 python tools/graph_debug.py $a 2>&1 | grep -v amdgpu.ids | tail -2
done
timeout 2400 python -m pytest tests -q -m gpu -x > $O/t_all.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error" $O/t_all.log | tail -n 8
bash tools/gpu_r03_h.sh
