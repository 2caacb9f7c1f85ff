#!/bin/bash
O=gpurun_out/r05j
mkdir -p $O
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_headline_parity_gpu.py tests/test_presplit_gpu.py tests/test_train_sample_gpu.py tests/test_ddp_gpu.py tests/test_f16_mode_gpu.py tests/test_config5_fullsize_gpu.py -q 2>&1 | tail -8 > $O/tests.log
for lds in 0 16384 24576 40960; do VD_GN_PS_LDS=$lds timeout 200 python tools/gn_ps_probe.py 2>/dev/null | grep -v amdgpu > $O/gn_probe_$lds.txt; done
cat $O/tests.log; tail -n 8 $O/gn_probe_0.txt; for lds in 16384 24576 40960; do tail -1 $O/gn_probe_$lds.txt; done
