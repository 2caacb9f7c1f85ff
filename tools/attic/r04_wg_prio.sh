#!/bin/bash
# experiment record: static priorities by wave-slot parity (-DVD_WG_STAMPS -DVD_WG_PRIO build of vd_gemm.hip linked into tools/diag/libvillan_hip_wgprio.so)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cp villandiffusion_amd/libvillan_hip.so /tmp/libvillan_hip.release.so
cp tools/diag/libvillan_hip_wgprio.so villandiffusion_amd/libvillan_hip.so
VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512 python tools/wg_stamps.py 32 > gpurun_out/wg_prio.txt 2>&1
VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512 python tools/wg_stamps.py 16 >> gpurun_out/wg_prio.txt 2>&1
for r in 1 2; do
VD_WGRAD_K32=1 VD_WGRAD_GROUP_TARGET=512 python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio k32', d['value'], d['ms_per_step'])" >> gpurun_out/wg_prio.txt
VD_WGRAD_K32=0 python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])" >> gpurun_out/wg_prio.txt
done
cp /tmp/libvillan_hip.release.so villandiffusion_amd/libvillan_hip.so
cat gpurun_out/wg_prio.txt
