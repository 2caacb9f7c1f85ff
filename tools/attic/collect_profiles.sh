#!/bin/bash
# Run on the GPU box (gpurun): the default bench line, the rocprofv3 kernel-time summary of the same command, and the two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) MI355X_MICROARCH.md prescribes.  Outputs under gpurun_out/r01/.
set -u
O=gpurun_out/r01
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_under_rocprof.json 2> $O/stats.err
if [ -n "${SKIP_PMC:-}" ]; then PMC_LIST=""; else PMC_LIST="FETCH_SIZE WRITE_SIZE"; fi
for c in $PMC_LIST; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 3 --warmup 2 --sample-images 0 --no-cpu --no-roofline > /dev/null 2> $O/pmc_$c.err
  python3 tools/pmc_summary.py $O/pmc_$c conv3_patch wgrad_patch bx3 conv3_pack gn_ gemm_plain adam slab_reduce > $O/pmc_$c.json
  rm -rf $O/pmc_$c
done
# MFMA utilisation: its own counter pass (no other tracing)
[ -n "${SKIP_PMC:-}" ] || timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 3 --warmup 2 --sample-images 0 --no-cpu --no-roofline > /dev/null 2> $O/pmc_mfma.err
[ -n "${SKIP_PMC:-}" ] || python3 tools/pmc_summary.py $O/pmc_mfma conv3_patch wgrad_patch bx3 gemm_plain gemm_kernel wgrad_kernel > $O/pmc_mfma.json
rm -rf $O/pmc_mfma
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $O/kernel_stats.csv; fi
rm -rf $O/stats
tail -1 $O/bench_default.json | cut -c1-400
head -12 $O/kernel_stats.csv
