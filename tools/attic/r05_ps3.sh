#!/bin/bash
O=gpurun_out/r05d
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_presplit_gpu.py -q -k "persistent_16x16x32_1x1 or groupnorm" 2>&1 | tail -8 > $O/tests.log
for r in 1 2; do
  VD_G32P_BM256=0 timeout 300 python tools/g32p_bm_ab.py > $O/g32p_bm0_$r.txt 2>&1
  VD_G32P_BM256=1 timeout 300 python tools/g32p_bm_ab.py > $O/g32p_bm1_$r.txt 2>&1
done
VD_G32P_BM256=0 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_bm0.json 2> $O/bench_bm0.err
VD_G32P_BM256=1 VD_BENCH_DETAIL=$O/detail_bm1.json timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > $O/bench_bm1.json 2> $O/bench_bm1.err
cat $O/tests.log; cat $O/g32p_bm0_2.txt $O/g32p_bm1_2.txt
for f in $O/bench_bm0.json $O/bench_bm1.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
