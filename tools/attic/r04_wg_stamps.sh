#!/bin/bash
# diagnostic build of the weight gradient with in-kernel stamps (tools/diag/libvillan_hip_wgstamps.so, -DVD_WG_STAMPS: never shipped; it replaces the
# release library only inside the GPU box's scratch copy of the tree).  Build it here first (the .so travels with gpurun, objects do not):
#   cd villandiffusion_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DVD_WG_STAMPS -c vd_gemm.hip -o /tmp/vd_gemm_wgstamps.o \
#     && mkdir -p ../../tools/diag && hipcc -shared -fPIC --offload-arch=gfx950 vd_api.o /tmp/vd_gemm_wgstamps.o vd_conv_k32p.o vd_gemm_k32p.o vd_norm.o vd_attn.o \
#        vd_attn_flash.o vd_elem.o -o ../../tools/diag/libvillan_hip_wgstamps.so
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cp villandiffusion_amd/libvillan_hip.so /tmp/libvillan_hip.release.so
cp tools/diag/libvillan_hip_wgstamps.so villandiffusion_amd/libvillan_hip.so
VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512 python tools/wg_stamps.py 32 > gpurun_out/wg_stamps.txt 2>&1
VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512 python tools/wg_stamps.py 16 >> gpurun_out/wg_stamps.txt 2>&1
for r in 1 2; do
VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512 python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k32', d['value'], d['ms_per_step'])" >> gpurun_out/wg_stamps.txt
VD_WGRAD_K32=0 python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])" >> gpurun_out/wg_stamps.txt
done
cp /tmp/libvillan_hip.release.so villandiffusion_amd/libvillan_hip.so
cat gpurun_out/wg_stamps.txt
