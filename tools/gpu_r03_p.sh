#!/bin/bash
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_ncsnpp.py tests/test_fid_gpu.py -q -m gpu -k "grouped or ncsnpp or hip_forward or measure_wiring" 2>&1 | tail -3
