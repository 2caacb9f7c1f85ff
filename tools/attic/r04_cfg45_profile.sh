#!/bin/bash
# round 4: kernel-level profile of BASELINE configs #4 / #5 (which kernels carry the step) + the opt-in act_out tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp

for c in ldm64 celebahq256; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 bench.py --config $c --steps 8 --warmup 3 > gpurun_out/prof_$c.log 2>&1
  f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/r04_${c}_kernel_stats.csv
done
head -5 gpurun_out/r04_ldm64_kernel_stats.csv | cut -c1-150
