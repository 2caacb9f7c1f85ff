#!/bin/bash
# round 6: the sampler runs 4 chunks on 4 streams -- does an UNSPLIT 8x8 / 4x4 convolution (fewer workgroups, no slab round trip, no epilogue launch)
# cost less of the chip than the split one?  Split targets through the planner's env knobs, same box, interleaved.
O=gpurun_out/r06
mkdir -p $O
rm -f $O/sampler_split_ab.txt
run() {
  env "$@" python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/sampler_split_ab.txt
}
for rep in 1 2; do
run VD_NOP=1
run VD_BX3_BIGSPLIT_TARGET=64
run VD_BX3_BIGSPLIT_TARGET=128
run VD_BX3_BIGSPLIT_TARGET=64 VD_BX3_SPLIT_TARGET=128
run VD_BX3_BIGSPLIT_TARGET=64 VD_BX3_SPLIT_TARGET=256
done
cat $O/sampler_split_ab.txt
