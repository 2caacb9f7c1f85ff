#!/bin/bash
run() { timeout 900 python bench.py --steps 30 --warmup 10 --no-exact --no-cpu --no-roofline --no-secondary --sample-images 128 --sample-streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('sample_ddpm1000_images_per_sec'))"; }
echo "== default"; run
echo "== PERSIST_MIN=512 SLOTS=256"; VD_GEMM_BX3_PERSIST_MIN=512 VD_GEMM_BX3_PERSIST_SLOTS=256 run
echo "== PERSIST_MIN=512 SLOTS=512"; VD_GEMM_BX3_PERSIST_MIN=512 VD_GEMM_BX3_PERSIST_SLOTS=512 run
echo "== PERSIST_MIN=256 SLOTS=256"; VD_GEMM_BX3_PERSIST_MIN=256 VD_GEMM_BX3_PERSIST_SLOTS=256 run
echo "== PERSIST_MIN=768 SLOTS=384"; VD_GEMM_BX3_PERSIST_MIN=768 VD_GEMM_BX3_PERSIST_SLOTS=384 run
echo "== default"; run
