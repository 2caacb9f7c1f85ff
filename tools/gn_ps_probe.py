"""GroupNorm producers at config #2's shapes (B = 128): f32 kernels vs pre-split kernels, forward and backward (pre-split only / both forms);
VD_GN_PS_LDS=<bytes> caps the occupancy of the pre-split kernels (experiment).   python tools/gn_ps_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops

DEV = torch.device("cuda")
B, G = 128, 32


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"f32 fwd": 0.0, "ps fwd": 0.0, "f32 bwd": 0.0, "ps bwd": 0.0, "ps bwd dual": 0.0}
for C, S in ((128, 32), (256, 32), (384, 32), (256, 16), (512, 16), (384, 16)):
    x = torch.randn(B, C, S, S, device=DEV)
    dy = torch.randn(B, C, S, S, device=DEV)
    ex = torch.randn(B, C, S, S, device=DEV)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    yp, dxp = ops.presplit_empty(x.shape, DEV), ops.presplit_empty(x.shape, DEV)
    m, r = torch.empty(B * G, device=DEV), torch.empty(B * G, device=DEV)
    wg, wb = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV)
    rs = torch.empty(B, C, device=DEV)
    t = {
        "f32 fwd": timed(lambda: ops.groupnorm_fwd(x, gamma, beta, y, m, r, G, 1e-6, True)),
        "ps fwd": timed(lambda: ops.groupnorm_fwd_presplit(x, gamma, beta, yp, m, r, G, 1e-6, True)),
        "f32 bwd": timed(lambda: ops.groupnorm_bwd(dy, x, m, r, gamma, beta, dx, wg, wb, G, True, extra=ex, rowsum=rs)),
        "ps bwd": timed(lambda: ops.groupnorm_bwd_presplit(dy, x, m, r, gamma, beta, None, dxp, wg, wb, G, True, rowsum=rs)),
        "ps bwd dual": timed(lambda: ops.groupnorm_bwd_presplit(dy, x, m, r, gamma, beta, dx, dxp, wg, wb, G, True, extra=ex, rowsum=rs)),
    }
    n = x.numel()
    byt = {"f32 fwd": 8, "ps fwd": 8, "f32 bwd": 16, "ps bwd": 12, "ps bwd dual": 20}
    print(f"C={C:3d} {S}x{S}: " + "  ".join(f"{k} {v:6.1f} us ({byt[k] * n / v / 1e6:.2f} TB/s)" for k, v in t.items()))
    for k, v in t.items():
        tot[k] += v
print(f"VD_GN_PS_LDS={os.environ.get('VD_GN_PS_LDS', '0')}: sums " + "  ".join(f"{k} {v:.0f} us" for k, v in tot.items()))
