#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "one_launch" 2>&1 | tail -6 > gpurun_out/gnchunk1b.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/gnchunk1b.txt
timeout 300 python bench.py --config celebahq256 --steps 8 --warmup 3 2>/dev/null | tail -1 | cut -c1-300 >> gpurun_out/gnchunk1b.txt
cat gpurun_out/gnchunk1b.txt
