#!/bin/bash
# diagnostic build of the persistent convolution with in-kernel stamps (tools/diag/libvillan_hip_stamps.so, built with -DVD_K32P_STAMPS: never shipped;
# it replaces the release library only inside the GPU box's scratch copy of the tree)
mkdir -p gpurun_out/r04
cp villandiffusion_amd/libvillan_hip.so /tmp/libvillan_hip.release.so
cp tools/diag/libvillan_hip_stamps.so villandiffusion_amd/libvillan_hip.so
python tools/k32p_stamps.py > gpurun_out/r04/k32p_stamps_reg.txt 2>&1
VD_K32P_DMA=1 python tools/k32p_stamps.py > gpurun_out/r04/k32p_stamps_dma.txt 2>&1
cp /tmp/libvillan_hip.release.so villandiffusion_amd/libvillan_hip.so
cat gpurun_out/r04/k32p_stamps_reg.txt; echo ---- DMA; cat gpurun_out/r04/k32p_stamps_dma.txt
