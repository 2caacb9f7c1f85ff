#!/bin/bash
# round 4: weight-gradient inner loop with hand-pipelined fragment reads (B) vs hipcc's schedule (A = tools/diag/libvillan_hip_wgrad_nopipe.so), same box
mkdir -p gpurun_out/r04
out=gpurun_out/r04/wgrad_pipe_ab.txt
: > $out
cp villandiffusion_amd/libvillan_hip.so /tmp/lib_B.so
cp tools/diag/libvillan_hip_wgrad_nopipe.so /tmp/lib_A.so
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "weight_gradient or grouped_weight" > gpurun_out/r04/t_wgrad.log 2>&1
tail -n 3 gpurun_out/r04/t_wgrad.log >> $out
for rep in 1 2 3; do
for v in A B; do
  cp /tmp/lib_$v.so villandiffusion_amd/libvillan_hip.so
  python bench.py --mode train --no-exact --no-cpu --steps 30 > /tmp/b.json 2>/dev/null
  python - "$v" >> $out <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
det = json.load(open("gpurun_out/bench_detail.json"))
rows = {k["kernel"]: k["ms"] for k in det["train_step_kernels"] if "wgrad" in k["kernel"]}
print(sys.argv[1], d["ms_per_step"], {k: v for k, v in rows.items()})
PY
done
done
cp /tmp/lib_B.so villandiffusion_amd/libvillan_hip.so
cat $out
