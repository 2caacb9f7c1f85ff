#!/bin/bash
# round 4, first GPU call: changed-code tests, the default bench line, the secondary config lines
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_fid_gpu.py tests/test_train_sample_gpu.py -q -m gpu -x > gpurun_out/r04/t_first.log 2>&1
tail -n 5 gpurun_out/r04/t_first.log
timeout 600 python bench.py > gpurun_out/r04/bench_default_0.json 2> gpurun_out/r04/bench_default_0.err
tail -c 1500 gpurun_out/r04/bench_default_0.json
cp gpurun_out/bench_detail.json gpurun_out/r04/bench_detail_0.json
timeout 600 python bench.py --config celebahq256 --steps 5 --warmup 2 > gpurun_out/r04/bench_cfg4_0.json 2> gpurun_out/r04/bench_cfg4_0.err
tail -c 600 gpurun_out/r04/bench_cfg4_0.json; grep " ms " gpurun_out/r04/bench_cfg4_0.err | head -14
timeout 600 python bench.py --config ldm64 --steps 5 --warmup 2 > gpurun_out/r04/bench_cfg5_0.json 2> gpurun_out/r04/bench_cfg5_0.err
tail -c 600 gpurun_out/r04/bench_cfg5_0.json; grep " ms " gpurun_out/r04/bench_cfg5_0.err | head -14
