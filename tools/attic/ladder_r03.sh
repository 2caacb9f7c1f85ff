#!/bin/bash
# Same-box ladder of the round-3 switches: everything off (about the round-2 configuration), then one more switch on per line, last line = shipped defaults.
# bench.py legs: training batch 128 (30 timed steps) and DDPM-1000 sampling of 384 images.  Output: img/s, ms/step, DDPM-1000 img/s.
run() { timeout 900 python bench.py --steps 30 --warmup 10 --no-exact --no-cpu --no-roofline --no-secondary --sample-images 384 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('sample_ddpm1000_images_per_sec'))"; }
export VD_BX3_K32_OFF=1 VD_W1X1_WIDE=0 VILLAN_FUSE_GN_BWD=0 VILLAN_PAGEABLE_H2D=1 VD_BENCH_HOST_POSITIONS=1 VD_BX3_S2_OFF=1 VD_GN_WAVE_OFF=1 VD_SPLITK_EPI_SCALAR=1 VILLAN_GN_STATS_IN_EPILOGUE=0
EXTRA="--sample-streams 1"
echo "all round-3 switches off (32x32x16 convolutions, 128-wide 1x1 weight gradient, separate rowsum / add_strided, pageable copies, f32 stride-2, block GroupNorm, scalar epilogue, statistics pass, 1 sampler stream):"; run
unset VD_BX3_K32_OFF; echo "+ 16x16x32 convolution kernel (conv3_k32_kernel):"; run
unset VD_W1X1_WIDE; echo "+ 256-wide 1x1 weight-gradient tiles:"; run
unset VILLAN_FUSE_GN_BWD; echo "+ row sums / skip adds in the GroupNorm backward:"; run
unset VILLAN_PAGEABLE_H2D VD_BENCH_HOST_POSITIONS; echo "+ resident batch positions (no blocking host->device copy):"; run
unset VD_BX3_S2_OFF; echo "+ stride-2 convolution on the split-precision kernel:"; run
unset VD_GN_WAVE_OFF; echo "+ wave-per-group GroupNorm at 8x8 / 4x4:"; run
unset VD_SPLITK_EPI_SCALAR; echo "+ float4 split-K epilogue:"; run
unset VILLAN_GN_STATS_IN_EPILOGUE; echo "+ GroupNorm statistics from conv1's epilogue (no-grad forward):"; run
EXTRA="--sample-streams 3"; echo "+ 3 concurrent sampler chunks (shipped defaults):"; run
