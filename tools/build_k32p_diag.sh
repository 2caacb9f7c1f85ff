#!/bin/bash
# Diagnostic builds of the persistent convolution (never shipped): tools/diag/libvillan_hip_k32p_var.so (-DVD_K32P_VARIANTS: ablation / scheduling
# flags through VD_K32P_FLAGS) and ..._stamps.so (+ -DVD_K32P_STAMPS: in-kernel s_memtime stamps).  Linked with the release objects of everything else.
set -e
cd "$(dirname "$0")/../villandiffusion_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/diag
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function"
hipcc $F -DVD_K32P_VARIANTS -c vd_conv_k32p.hip -o ../../tools/diag/k32p_var.o &
hipcc $F -DVD_K32P_VARIANTS -DVD_K32P_STAMPS -c vd_conv_k32p.hip -o ../../tools/diag/k32p_stamps.o &
hipcc $F -DVD_G32P_VARIANTS -c vd_gemm_k32p.hip -o ../../tools/diag/g32p_var.o &
hipcc $F -DVD_SM_STAMPS -c vd_conv_sm.hip -o ../../tools/diag/sm_stamps.o &
wait
OTHERS1=$(ls *.o | grep -v '^vd_gemm_k32p.o$')
hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS1 ../../tools/diag/g32p_var.o -o ../../tools/diag/libvillan_hip_g32p_var.so      # 1x1 kernel: VD_G32P_FLAGS ablations
OTHERS2=$(ls *.o | grep -v '^vd_conv_sm.o$')
hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS2 ../../tools/diag/sm_stamps.o -o ../../tools/diag/libvillan_hip_sm_stamps.so      # whole-K 8x8 / 4x4 kernel: per-wave segment sums
OTHERS=$(ls *.o | grep -v '^vd_conv_k32p.o$')
hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS ../../tools/diag/k32p_var.o -o ../../tools/diag/libvillan_hip_k32p_var.so
hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS ../../tools/diag/k32p_stamps.o -o ../../tools/diag/libvillan_hip_k32p_stamps.so
ls -la ../../tools/diag/*.so
