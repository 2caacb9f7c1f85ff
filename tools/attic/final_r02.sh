#!/bin/bash
# Round-2 closing run on the GPU box: the whole GPU test suite in chunks (own log + limit each), then the evidence collection.
source tools/gpu_tests_r02.sh
run t_kernels 600 tests/test_hip_kernels.py tests/test_cabi.py tests/test_vqmodel.py tests/test_ncsnpp.py
run t_unet 600 tests/test_unet_gpu.py tests/test_ddp_gpu.py
run t_train 1200 tests/test_train_sample_gpu.py
run t_headline 1500 tests/test_headline_parity_gpu.py
run t_cfg5 900 tests/test_config5_fullsize_gpu.py
run t_rest 900 tests --ignore=tests/test_hip_kernels.py --ignore=tests/test_cabi.py --ignore=tests/test_vqmodel.py --ignore=tests/test_ncsnpp.py --ignore=tests/test_unet_gpu.py --ignore=tests/test_ddp_gpu.py --ignore=tests/test_train_sample_gpu.py --ignore=tests/test_headline_parity_gpu.py --ignore=tests/test_config5_fullsize_gpu.py
cat gpurun_out/r02/summary.log
bash tools/collect_profiles_r02.sh
