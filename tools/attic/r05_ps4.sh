#!/bin/bash
O=gpurun_out/r05e
mkdir -p $O
timeout 1500 python -m pytest tests/test_presplit_gpu.py tests/test_unet_gpu.py -q -x -k "presplit or weight_gradient or backward_matches or batch128 or network" 2>&1 | tail -15 > $O/tests.log
VILLAN_PRESPLIT=0 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_off.json 2> $O/bench_off.err
VD_BENCH_DETAIL=$O/detail_on.json timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_on.json 2> $O/bench_on.err
VILLAN_PRESPLIT=0 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_off2.json 2> $O/bench_off2.err
timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_on2.json 2> $O/bench_on2.err
cat $O/tests.log
for f in $O/bench_off.json $O/bench_on.json $O/bench_off2.json $O/bench_on2.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
