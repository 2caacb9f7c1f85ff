#!/bin/bash
echo "== wide off"; VD_W1X1_WIDE=0 timeout 600 python -m pytest tests/test_ncsnpp.py -q -m gpu -x 2>&1 | tail -2
echo "== wide on, sync debug"; VD_SYNC_DEBUG=1 timeout 600 python -m pytest tests/test_ncsnpp.py -q -m gpu -x -s -k "forward_backward" 2>&1 | grep -v amdgpu.ids | grep -E "\[vd\]|fault|Fatal|passed|failed|HSA" | tail -8
