#!/bin/bash
# round 5, second pre-split run: GroupNorm backward producer + the network-level path, then the bench (default = pre-split) and its A/B (VILLAN_PRESPLIT=0)
O=gpurun_out/r05c
mkdir -p $O
timeout 1500 python -m pytest tests/test_presplit_gpu.py -q -s -k "backward or network or loudly" 2>&1 | tail -60 > $O/tests.log
for i in 1 2; do
  VILLAN_PRESPLIT=0 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_off_$i.json 2> $O/bench_off_$i.err
  VD_BENCH_DETAIL=$O/detail_on_$i.json timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_on_$i.json 2> $O/bench_on_$i.err
done
grep -E "passed|failed|^FAILED|parity" $O/tests.log | tail -30
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("host_submit_ms_per_step"), d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
    for k in d["top_kernels"]: print("   ", k["kernel"], k["launches"], k["ms"])
except Exception as e:
    print("ERR", e)
PY
done
tail -5 $O/bench_on_1.err
