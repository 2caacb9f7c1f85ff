#!/bin/bash
O=gpurun_out/r03p
mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu -x > gpurun_out/r03/t_all.log 2>&1
grep -E "Fatal|FAILED|passed|failed|Error" gpurun_out/r03/t_all.log | tail -n 6
VD_BENCH_DETAIL=$O/bench_detail.json timeout 1200 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
