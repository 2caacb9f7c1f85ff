#!/bin/bash
# round 4: whole GPU suite, then the evidence collection (bench line + rocprofv3 stats + PMC passes), then the secondary config lines
mkdir -p gpurun_out/r04 gpurun_out/r04p
timeout 2700 python -m pytest tests -q -m gpu -x > gpurun_out/r04/t_all.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error" gpurun_out/r04/t_all.log | tail -n 6
bash tools/collect_profiles_r04.sh > gpurun_out/r04p/collect.log 2>&1
tail -n 3 gpurun_out/r04p/collect.log | cut -c1-300
timeout 600 python bench.py --config celebahq256 --steps 5 --warmup 2 > gpurun_out/r04p/bench_cfg4.json 2> gpurun_out/r04p/bench_cfg4.err
timeout 600 python bench.py --config ldm64 --steps 5 --warmup 2 > gpurun_out/r04p/bench_cfg5.json 2> gpurun_out/r04p/bench_cfg5.err
tail -c 300 gpurun_out/r04p/bench_cfg4.json; tail -c 300 gpurun_out/r04p/bench_cfg5.json
