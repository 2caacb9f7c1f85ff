#!/bin/bash
mkdir -p gpurun_out/r03
run() {
VD_BENCH_DETAIL=gpurun_out/r03/q_detail.json timeout 900 python bench.py --steps 30 --warmup 10 --no-exact --no-cpu --no-secondary --sample-images 128 --sample-streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('sample_ddpm1000_images_per_sec'))"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/q_detail.json'))
out=[]
for tab in ('train_step_kernels','sampler_step_kernels'):
    for r in d[tab]:
        if 'k32' in r['kernel'] and r['launches']>1: out.append(f"{r['kernel'][17:]} {r['avg_us']:.1f}")
print(' | '.join(out))
PY
}
for i in 1 2; do
echo "== DPP (default)"; run
echo "== VD_BX3_K32_FLAGS=16 (position 1 from LDS)"; VD_BX3_K32_FLAGS=16 run
done
