#!/bin/bash
# round 4: training forward with folded GroupNorm (statistics pass + loader) and silu(gn(.)) recomputed on the weight-gradient side stream -- parity + A/B
mkdir -p gpurun_out/r04
out=gpurun_out/r04/gn_defer_ab.txt
: > $out
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_headline_parity_gpu.py -q -m gpu -x > gpurun_out/r04/t_gn_defer.log 2>&1
tail -n 3 gpurun_out/r04/t_gn_defer.log >> $out
for rep in 1 2 3; do
for cfg in "VILLAN_DEFER_GN_FWD=0" "VILLAN_DEFER_GN_FWD=1"; do
    r=$(env $cfg python bench.py --mode train --no-exact --no-cpu --no-roofline --steps 30 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$cfg train $r" >> $out
done
done
cat $out
