"""Split-precision (bf16x3) 3x3 convolution vs the exact-f32 MFMA kernel: error and time per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops

torch.manual_seed(0)
dev = "cuda"
SHAPES = [  # (B, C, M, S_out, mode)
    (128, 128, 128, 32, ops.B_CONV3), (128, 128, 128, 32, ops.B_CONV3_T), (128, 256, 128, 32, ops.B_CONV3), (128, 384, 128, 32, ops.B_CONV3),
    (128, 256, 256, 16, ops.B_CONV3), (128, 256, 256, 16, ops.B_CONV3_T), (128, 512, 256, 16, ops.B_CONV3), (128, 128, 256, 16, ops.B_CONV3),
    (128, 256, 256, 8, ops.B_CONV3), (128, 512, 256, 8, ops.B_CONV3), (128, 256, 256, 8, ops.B_CONV3_T),
    (128, 128, 128, 32, ops.B_CONV3_UP), (128, 256, 256, 16, ops.B_CONV3_UP), (100, 128, 192, 32, ops.B_CONV3),
]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]


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


for (B, Cc, M, S, mode) in SHAPES:
    Sin = S // 2 if mode == ops.B_CONV3_UP else S
    x = torch.randn(B, Cc, Sin, Sin, device=dev)
    w = torch.randn(M, Cc * 9, device=dev) / (Cc * 9) ** 0.5
    bias = torch.randn(M, device=dev)
    res = torch.randn(B, M, S, S, device=dev)
    temb = torch.randn(B, M, device=dev)
    o0 = torch.empty(B, M, S, S, device=dev)
    o1 = torch.empty_like(o0)
    pk = ops.conv3_pack_weights(w, M, Cc)
    f0 = lambda: ops.conv3x3(x, w, bias, o0, mode=mode, rowadd=temb, rowadd_bstride=M, residual=res)
    f1 = lambda: ops.conv3x3(x, w, bias, o1, mode=mode, rowadd=temb, rowadd_bstride=M, residual=res, a_packed=pk)
    f0(); f1()
    torch.cuda.synchronize()
    err = (o1 - o0).abs().max().item() / o0.std().item()
    t0, t1 = timeit(f0), timeit(f1)
    tp = timeit(lambda: ops.conv3_pack_weights(w, M, Cc, out=pk))
    fl = 2.0 * M * Cc * 9 * B * S * S
    print(f"B={B} C={Cc} M={M} S={S} mode={mode}: err/std {err:.2e}  f32 {t0:7.1f} us {fl / t0 / 1e6:6.1f} TF | bx3 {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF | x{t0 / t1:.2f} | pack {tp:.1f} us",
          flush=True)
