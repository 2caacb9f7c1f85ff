#!/bin/bash
O=gpurun_out/r05i
mkdir -p $O
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_headline_parity_gpu.py tests/test_presplit_gpu.py -q -x -k "not fullsize" 2>&1 | tail -6 > $O/tests.log
for i in 1 2; do
  VILLAN_SC_STREAM=0 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline > $O/bench_sc0_$i.json 2> $O/bench_sc0_$i.err
  VILLAN_SC_STREAM=1 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline > $O/bench_sc1_$i.json 2> $O/bench_sc1_$i.err
done
cat $O/tests.log
for f in $O/bench_sc*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
