#!/bin/bash
# round 6: every sampler chunk stream on its own quarter of the CUs (hipExtStreamCreateWithCUMask), same box, interleaved
O=gpurun_out/r06
mkdir -p $O
rm -f $O/sampler_cumask_ab.txt
sa() {
  env "$@" timeout 300 python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>$O/sampler_cumask_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $*', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/sampler_cumask_ab.txt
}
for rep in 1 2; do
sa VD_NOP=1
sa VILLAN_SAMPLER_CU_MASK=contiguous
sa VILLAN_SAMPLER_CU_MASK=interleaved
done
cat $O/sampler_cumask_ab.txt; tail -3 $O/sampler_cumask_err.txt
