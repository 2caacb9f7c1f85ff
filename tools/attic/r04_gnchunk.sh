#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/gn_chunk_ab.txt
: > $out
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_unet_gpu.py tests/test_config5_fullsize_gpu.py -q -m gpu -x -k "groupnorm or celebahq or ldm or config5 or fullsize" > gpurun_out/r04/t_gnchunk.log 2>&1
tail -n 3 gpurun_out/r04/t_gnchunk.log >> $out
for rep in 1 2; do
for cfg in "VD_GN_CHUNK_MB=0" "VD_GN_CHUNK_MB=64" "VD_GN_CHUNK_MB=128"; do
  env $cfg python bench.py --config celebahq256 --steps 5 --warmup 2 > /tmp/c4.json 2> /tmp/c4.err
  python - "$cfg" >> $out <<'PY'
import json, sys
d = json.loads(open("/tmp/c4.json").read().strip().splitlines()[-1])
rows = {k["kernel"]: k["ms"] for k in d["top_kernels"] if "groupnorm" in k["kernel"]}
print(sys.argv[1], d["ms_per_step"], rows)
PY
done
done
cat $out
