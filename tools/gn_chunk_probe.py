"""Achieved HBM rate of the chunked GroupNorm kernels (groups > 28672 elements) per shape, one-launch form vs two-launch form
(VD_GN_CHUNK1_OFF=1 selects the latter for the whole process).   python tools/gn_chunk_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from villandiffusion_amd import ops  # noqa: E402

DEV = torch.device("cuda")
for (B, C, H) in [(8, 128, 256), (8, 256, 256), (8, 128, 128), (8, 256, 128), (8, 256, 64), (8, 512, 64)]:
    x = torch.randn(B, C, H, H, device=DEV)
    dy = torch.randn(B, C, H, H, device=DEV)
    ex = torch.randn(B, C, H, H, device=DEV)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
    wg, wb, rs = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV), torch.empty(B, C, device=DEV)

    def fwd():
        ops.groupnorm_fwd(x, gamma, beta, y, mean, rstd, 32, 1e-6, True)

    def bwd():
        ops.groupnorm_bwd(dy, x, mean, rstd, gamma, beta, dx, wg, wb, 32, True, extra=ex, rowsum=rs, rowsum_ld=C)

    res = []
    for fn, sweeps in ((fwd, 2), (bwd, 4)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res.append(f"{us:8.1f} us {sweeps * x.numel() * 4 / us / 1e3:7.1f} GB/s (algorithmic {sweeps} sweeps)")
    print(f"groupnorm {C:4d} ch @{H}^2 B={B} (group {C // 32 * H * H} elements): fwd {res[0]} | bwd+extra+rowsum {res[1]}")
