#!/bin/bash
# Run on the GPU box (gpurun): round-6 evidence.  Every rocprofv3 summary covers ONE dispatch population (round 6: --no-ddp-path on every
# profiled training leg -- the multi-rank-schedule leg of the default line used to run under the profiler too and mixed two schedules plus RCCL kernels):
#   bench.py --mode train --serial-wgrad (weight gradients on the launch stream: every kernel's duration is its own)
#                           -> r06_train_kernel_stats.csv  (+ the FETCH_SIZE / WRITE_SIZE / MFMA PMC passes, each its own run, --kernel-trace only)
#   bench.py --mode sample  -> r06_sample_kernel_stats.csv
# plus the default bench line.  Outputs under gpurun_out/r06p/; tools/update_profiles_r06.py copies the summaries into profiles/.
set -u
O=gpurun_out/r06p
mkdir -p $O
export TMPDIR=/tmp
VD_BENCH_DETAIL=$O/bench_detail.json timeout 1200 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 bench.py --mode train --serial-wgrad --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path > $O/bench_train_under_rocprof.json 2> $O/stats_train.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sample -- python3 bench.py --mode sample --no-cpu --no-f16 --no-roofline --no-secondary --sample-images 128 > $O/bench_sample_under_rocprof.json 2> $O/stats_sample.err
for w in train sample; do
  f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/${w}_kernel_stats.csv; fi
  rm -rf $O/stats_$w
done
if [ -z "${SKIP_PMC:-}" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --mode train --serial-wgrad --steps 3 --warmup 2 --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path > /dev/null 2> $O/pmc_$c.err
    python3 tools/pmc_summary.py $O/pmc_$c > $O/pmc_$c.json
    rm -rf $O/pmc_$c
  done
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --mode train --serial-wgrad --steps 3 --warmup 2 --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path > /dev/null 2> $O/pmc_mfma.err
  python3 tools/pmc_summary.py $O/pmc_mfma > $O/pmc_mfma.json
  rm -rf $O/pmc_mfma
fi
if [ -z "${SKIP_PMC:-}" ]; then          # the sampler's dispatches: traffic and MFMA counters of a 30-step DDPM loop (eager launches: one dispatch per kernel)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcs_$c -- python3 bench.py --mode sample --sample-steps 30 --sample-images 128 --no-cpu --no-f16 --no-roofline --no-secondary > /dev/null 2> $O/pmcs_$c.err
    python3 tools/pmc_summary.py $O/pmcs_$c > $O/pmc_sample_$c.json
    rm -rf $O/pmcs_$c
  done
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcs_mfma -- python3 bench.py --mode sample --sample-steps 30 --sample-images 128 --no-cpu --no-f16 --no-roofline --no-secondary > /dev/null 2> $O/pmcs_mfma.err
  python3 tools/pmc_summary.py $O/pmcs_mfma > $O/pmc_sample_mfma.json
  rm -rf $O/pmcs_mfma
fi
python3 tools/shape_probe.py > $O/shape_probe.txt 2>&1
# sustained matrix-pipe rates from registers / from LDS / with random operand bits (hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o build/mfma_peak)
if [ -x tools/mfma_peak ]; then timeout 300 tools/mfma_peak > $O/mfma_sustained.txt 2>&1; fi
if [ -z "${SKIP_PMC:-}" ]; then
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_wait -- python3 bench.py --mode train --serial-wgrad --steps 3 --warmup 2 --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path > /dev/null 2> $O/pmc_wait.err
  python3 tools/pmc_summary.py $O/pmc_wait > $O/pmc_wait.json
  rm -rf $O/pmc_wait
fi
# BASELINE configs #4 / #5: secondary bench lines + their kernel summaries (rocprofv3 --kernel-trace --stats over the same command)
# Round 5: each secondary configuration also gets its FETCH_SIZE / WRITE_SIZE / MFMA PMC passes (separate runs, --kernel-trace only), so that its
# line carries roofline.traffic and a traffic_over_algorithmic table like config #2's (tools/update_profiles_r06.py -> r06_pmc_traffic_cfg4/5.json).
for c in celebahq256 ldm64; do
  VD_BENCH_DETAIL=$O/bench_detail_$c.json timeout 600 python3 bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err
  if [ -z "${SKIP_PMC:-}" ]; then
    for k in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $O/pmc_${c}_$k -- python3 bench.py --config $c --steps 2 --warmup 2 --serial-wgrad --no-roofline > /dev/null 2> $O/pmc_${c}_$k.err
      python3 tools/pmc_summary.py $O/pmc_${c}_$k > $O/pmc_${c}_$k.json
      rm -rf $O/pmc_${c}_$k
    done
    timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${c}_mfma -- python3 bench.py --config $c --steps 2 --warmup 2 --serial-wgrad --no-roofline > /dev/null 2> $O/pmc_${c}_mfma.err
    python3 tools/pmc_summary.py $O/pmc_${c}_mfma > $O/pmc_${c}_mfma.json
    rm -rf $O/pmc_${c}_mfma
  fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 bench.py --config $c --steps 8 --warmup 3 --serial-wgrad --no-roofline > /dev/null 2> $O/stats_$c.err
  f=$(find $O/stats_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/${c}_kernel_stats.csv; fi
  rm -rf $O/stats_$c
done
tail -1 $O/bench_default.json | cut -c1-600
head -8 $O/train_kernel_stats.csv | cut -c1-160
head -8 $O/sample_kernel_stats.csv | cut -c1-160
# CPU-oracle thread sweep (review Weak 9): the B = 128 oracle training step at 16 .. 256 host threads
if [ -z "${SKIP_CPU_SWEEP:-}" ]; then timeout 1500 python3 tools/cpu_thread_sweep.py > $O/cpu_threads.txt 2>&1; fi
