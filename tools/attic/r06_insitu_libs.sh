#!/bin/bash
# round 6: changes that were shipped on probe evidence, judged inside the step: the library with the OLD kernel of each against the shipped one, same box, interleaved
O=gpurun_out/r06
mkdir -p $O
rm -f $O/insitu_imgs2_ab.txt
cp villandiffusion_amd/libvillan_hip.so /tmp/lib_shipped.so
tr() {
  timeout 300 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train $1', d['value'], d['ms_per_step'])" >> $O/insitu_imgs2_ab.txt
}
sa() {
  timeout 300 python bench.py --mode sample --sample-steps 150 --sample-images 1024 --no-cpu --no-f16 --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample $1', d['sample_ddpm1000_images_per_sec'], d['sample_seconds'])" >> $O/insitu_imgs2_ab.txt
}
for rep in 1 2 3; do
for v in shipped imgs2; do
  if [ $v = shipped ]; then cp /tmp/lib_shipped.so villandiffusion_amd/libvillan_hip.so; else cp tools/diag/libvillan_hip_$v.so villandiffusion_amd/libvillan_hip.so; fi
  tr $v
  sa $v
done
done
cp /tmp/lib_shipped.so villandiffusion_amd/libvillan_hip.so
cat $O/insitu_imgs2_ab.txt
