#!/bin/bash
# round 6: converting loader of conv3_k32p_kernel with 16-byte row loads (shipped) against round 5's dword loads (VD_K32P_ROWLD=0): f32 inputs, modes 0 / 1 and
# the GroupNorm-folding mode 3 (the sampler's), same box, interleaved; the sha1 column must not move
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2; do
VD_K32P_ROWLD=0 python tools/k32p_probe.py --f32 > $O/rowld_off_$rep.txt 2>&1
python tools/k32p_probe.py --f32 > $O/rowld_on_$rep.txt 2>&1
done
paste <(cut -c1-76 $O/rowld_off_1.txt) <(cut -c24-76 $O/rowld_on_1.txt) | grep -v amdgpu
tail -qn1 $O/rowld_off_1.txt $O/rowld_on_1.txt $O/rowld_off_2.txt $O/rowld_on_2.txt
