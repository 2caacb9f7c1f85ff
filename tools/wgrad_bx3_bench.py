"""Split-precision (bf16x3) 3x3 weight gradient vs the exact-f32 MFMA kernel: error and time per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops

torch.manual_seed(0)
dev = "cuda"
SHAPES = [(128, 128, 128, 32), (128, 256, 128, 32), (128, 384, 128, 32), (128, 256, 256, 16), (128, 512, 256, 16), (128, 128, 256, 16),
          (128, 256, 256, 8), (128, 512, 256, 8), (100, 192, 128, 32), (3, 64, 64, 16)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, Cc, M, S) in SHAPES:
    x = torch.randn(B, Cc, S, S, device=dev)
    dy = torch.randn(B, M, S, S, device=dev)
    need = max(ops.wgrad_ws_floats(M, Cc, 9, B, S * S, mode=ops.B_CONV3), ops.wgrad_ws_floats(M, Cc, 9, B, S * S, mode=ops.B_CONV3, math_mode=1))
    ws = torch.empty(max(need, 4), device=dev)
    g0 = torch.zeros(M, Cc * 9, device=dev)
    g1 = torch.zeros_like(g0)
    f0 = lambda: ops.conv_wgrad(dy, x, g0, ops.B_CONV3, ws)
    f1 = lambda: ops.conv_wgrad(dy, x, g1, ops.B_CONV3, ws, math_mode=1)
    f0(); f1()
    torch.cuda.synchronize()
    err = (g1 - g0).abs().max().item() / g0.std().item()
    t0, t1 = timeit(f0), timeit(f1)
    fl = 2.0 * M * Cc * 9 * B * S * S
    print(f"B={B} C={Cc} M={M} S={S}: err/std {err:.2e}  f32 {t0:7.1f} us {fl / t0 / 1e6:6.1f} TF | bx3 {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF | x{t0 / t1:.2f}", flush=True)
