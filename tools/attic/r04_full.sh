#!/bin/bash
# round 4 checkpoint: the whole GPU suite + the default bench line + the secondary config lines
mkdir -p gpurun_out/r04
timeout 2700 python -m pytest tests -q -m gpu -x > gpurun_out/r04/t_all.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error" gpurun_out/r04/t_all.log | tail -n 6
timeout 900 python bench.py > gpurun_out/r04/bench_default_1.json 2> gpurun_out/r04/bench_default_1.err
tail -c 1200 gpurun_out/r04/bench_default_1.json; grep -E "train  |sampler" gpurun_out/r04/bench_default_1.err | head -30
cp gpurun_out/bench_detail.json gpurun_out/r04/bench_detail_1.json
timeout 600 python bench.py --config celebahq256 --steps 5 --warmup 2 > gpurun_out/r04/bench_cfg4_1.json 2> gpurun_out/r04/bench_cfg4_1.err
tail -c 400 gpurun_out/r04/bench_cfg4_1.json; grep " ms " gpurun_out/r04/bench_cfg4_1.err | head -12
timeout 600 python bench.py --config ldm64 --steps 5 --warmup 2 > gpurun_out/r04/bench_cfg5_1.json 2> gpurun_out/r04/bench_cfg5_1.err
tail -c 400 gpurun_out/r04/bench_cfg5_1.json; grep " ms " gpurun_out/r04/bench_cfg5_1.err | head -12
