"""Tile / split sweep of the generic weight-gradient kernel on the shapes the CIFAR10 step runs through it.
   python tools/wgrad_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3, B_PLAIN
from tools.gemm_bench_util import timeit

B = 128
CASES = [("1x1", 384, 128, 32), ("1x1", 256, 768, 16), ("1x1", 256, 256, 16), ("1x1", 512, 256, 16), ("1x1", 768, 256, 16),
         ("1x1", 512, 256, 8), ("3x3", 256, 256, 8), ("3x3", 512, 256, 8), ("3x3", 256, 256, 4), ("3x3", 512, 256, 4)]
for kind, cin, cout, H in CASES:
    T = 1 if kind == "1x1" else 9
    mode = B_PLAIN if T == 1 else B_CONV3
    x = torch.randn(B, cin, H, H, device="cuda")
    dy = torch.randn(B, cout, H, H, device="cuda")
    dw = torch.empty(cout, cin * T, device="cuda")
    flops = 2.0 * cout * cin * T * B * H * H
    ws = torch.empty(256 * cout * cin * T // (4 if T == 9 else 1) + 4, device="cuda")
    res = []
    for tile in ((0, 1, 3) if T == 1 else (0, 1, 3, 4)):
        for splits in ((0,) if tile in (0, 4) else (0, 16, 32, 64, 128)):
            try:
                if splits * cout * cin * T > ws.numel():
                    continue
                ms = timeit(lambda: ops.conv_wgrad(dy, x, dw, mode, ws, splits=splits, tile=tile))
                res.append(f"t{tile}/s{splits}:{flops / ms / 1e9:5.1f}TF({ms * 1e3:.0f}us)")
            except Exception as e:
                res.append(f"t{tile}/s{splits}:fail")
    print(f"{kind} {cin}->{cout}@{H}: " + " ".join(res))
