#!/bin/bash
O=gpurun_out/r05n
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_kernels.py -q -k "split_precision_conv3x3_forward_and_dgrad or tile_choice or persistent_16x16x32_convolution" 2>&1 | tail -8 > $O/tests.log
cat $O/tests.log
for i in 1 2; do
  VD_K32P8_OFF=1 timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline > $O/bench_k8off_$i.json 2> $O/bench_k8off_$i.err
  timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline > $O/bench_k8on_$i.json 2> $O/bench_k8on_$i.err
done
VD_BENCH_DETAIL=$O/detail.json timeout 900 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path > /dev/null 2> $O/bench_detail.err
for f in $O/bench_k8*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05n/detail.json"))
for k in d["train_step_kernels"]:
    if "<8," in k["kernel"] or "<4," in k["kernel"]: print(k["ms"], k["launches"], k["avg_us"], k["kernel"])
PY
