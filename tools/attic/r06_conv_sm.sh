#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
for rep in 1 2; do
VD_CONV_SM_OFF=1 python tools/conv_sm_probe.py > $O/conv_sm_off_$rep.txt 2>&1
VD_CONV_SM_IMG1=1 python tools/conv_sm_probe.py > $O/conv_sm_img1_$rep.txt 2>&1
python tools/conv_sm_probe.py > $O/conv_sm_on_$rep.txt 2>&1
done
paste <(cut -c1-55 $O/conv_sm_off_1.txt) <(cut -c24-55 $O/conv_sm_img1_1.txt) <(cut -c24-55 $O/conv_sm_on_1.txt) | grep -v amdgpu
tail -qn1 $O/conv_sm_off_2.txt $O/conv_sm_img1_2.txt $O/conv_sm_on_2.txt
