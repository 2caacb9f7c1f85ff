#!/bin/bash

for LIBV in villandiffusion_amd/libvillan_hip.so tools/diag/libvillan_hip_noxch.so villandiffusion_amd/libvillan_hip.so tools/diag/libvillan_hip_noxch.so; do FEWOUT_LIB=$LIBV python - <<'PY'
import torch, math, os
from villandiffusion_amd import lib as _L
_L.LIB_PATH = os.path.abspath(os.environ['FEWOUT_LIB'])
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3
for (B, cin, cout, S) in [(128, 128, 3, 32), (8, 128, 3, 256)]:
    x = torch.randn(B, cin, S, S, device="cuda"); w = torch.randn(cout, cin * 9, device="cuda") / math.sqrt(cin * 9)
    out = torch.empty(B, cout, S, S, device="cuda"); bias = torch.randn(cout, device="cuda")
    for _ in range(5): ops.conv3x3(x, w, bias, out, mode=B_CONV3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.conv3x3(x, w, bias, out, mode=B_CONV3)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    ref = torch.nn.functional.conv2d(x, w.view(cout, cin, 3, 3), bias, padding=1)
    print(f"conv_out {cin}->{cout} @{S} B={B}: {us:.1f} us  {4 * x.numel() / us / 1e6:.2f} TB/s  tile {ops.LAST_GEMM_TILE}  max err {float((out - ref).abs().max()):.2e}")
PY
done
