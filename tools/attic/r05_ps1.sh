#!/bin/bash
# round 5, first pre-split run: kernel-level parity + the same-box A/B of the weight gradient / GroupNorm producers
O=gpurun_out/r05b
mkdir -p $O
timeout 1500 python -m pytest tests/test_presplit_gpu.py tests/test_ddp_gpu.py -q -s -k "not bench_gpus" 2>&1 | tail -70 > $O/tests.log
timeout 600 python tools/wgrad_ps_ab.py > $O/wgrad_ps_ab.txt 2>&1
tail -c 2500 $O/tests.log
cat $O/wgrad_ps_ab.txt | tail -30
