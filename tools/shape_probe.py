"""Per-SHAPE timing (HIP events, 30 back-to-back launches) of the kernels of the config #2 training step that are launched at many
shapes: the split-precision 1x1 convolutions (forward and input gradient), GroupNorm forward / backward and the fused attention core.
Prints algorithmic TFLOP/s and GB/s (operands read once, result written once) so the binding roofline can be read per shape.
    python tools/shape_probe.py [gemm] [gn] [attn] [conv3]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.gemm_bench_util import timeit  # noqa: E402
from villandiffusion_amd import ops  # noqa: E402

B = 128
what = set(sys.argv[1:]) or {"gemm", "gn", "attn"}
if "gemm" in what:
    # (cin, cout, H, residual): shortcuts of the has-shortcut resnets and the attention projections (qkv 256 -> 768, out 256 -> 256)
    shapes = [(128, 256, 16, False), (512, 256, 4, False), (512, 256, 8, False), (512, 256, 16, False), (384, 256, 16, False),
              (384, 128, 32, False), (256, 128, 32, False), (256, 768, 16, False), (256, 256, 16, True), (768, 256, 16, False),
              (256, 512, 16, False), (128, 256, 32, False), (128, 384, 32, False)]
    for cin, cout, H, res in shapes:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, device="cuda") / math.sqrt(cin)
        bias = torch.randn(cout, device="cuda")
        out = torch.empty(B, cout, H, H, device="cuda")
        r = torch.randn(B, cout, H, H, device="cuda") if res else None
        pk = ops.conv3_pack_weights(w, cout, cin, taps=1)
        ms = timeit(lambda: ops.conv1x1(x, w, bias, out, residual=r, a_packed=pk), n=30)
        flops = 2.0 * cout * cin * B * H * H
        nbytes = 4.0 * B * H * H * (cin + cout * (2 if res else 1))
        print(f"conv1x1 {cin:4d}->{cout:4d} @{H:2d}^2 res={int(res)}: {ms * 1e3:7.1f} us  {flops / ms / 1e9:6.1f} TF  {nbytes / ms / 1e6:7.1f} GB/s  "
              f"tiles {((cout + 127) // 128) * (B * H * H // 128)}")
if "gn" in what:
    for ch, H in [(128, 32), (256, 32), (384, 32), (256, 16), (512, 16), (384, 16), (256, 8), (512, 8), (256, 4), (512, 4)]:
        x = torch.randn(B, ch, H, H, device="cuda")
        y = torch.empty_like(x)
        g_, b_ = torch.ones(ch, device="cuda"), torch.zeros(ch, device="cuda")
        mean, rstd = torch.empty(B * 32, device="cuda"), torch.empty(B * 32, device="cuda")
        ms = timeit(lambda: ops.groupnorm_fwd(x, g_, b_, y, mean, rstd, 32, 1e-6, True), n=30)
        dy, dx = torch.randn_like(x), torch.empty_like(x)
        wg, wb = torch.empty(B * ch, device="cuda"), torch.empty(B * ch, device="cuda")
        ms2 = timeit(lambda: ops.groupnorm_bwd(dy, x, mean, rstd, g_, b_, dx, wg, wb, 32, True), n=30)
        ms3 = timeit(lambda: ops.groupnorm_bwd(dy, x, mean, rstd, g_, b_, dx, wg, wb, 32, True, extra=y), n=30)
        n = x.numel()
        print(f"groupnorm {ch:4d} @{H:2d}^2: fwd {ms * 1e3:6.1f} us {8.0 * n / ms / 1e6:7.1f} GB/s | bwd {ms2 * 1e3:6.1f} us {12.0 * n / ms2 / 1e6:7.1f} GB/s | "
              f"bwd+extra {ms3 * 1e3:6.1f} us {16.0 * n / ms3 / 1e6:7.1f} GB/s")
if "attn" in what:
    for heads, d in [(1, 256), (1, 512), (8, 32)]:
        C, N = heads * d, 256
        qkv = torch.randn(B, 3 * C, N, device="cuda")
        o, P = torch.empty(B, C, N, device="cuda"), torch.empty(B, heads, N, N, device="cuda")
        do, dS, dqkv = torch.randn(B, C, N, device="cuda"), torch.empty(B, heads, N, N, device="cuda"), torch.empty(B, 3 * C, N, device="cuda")
        sc = 1 / math.sqrt(d)
        fl = 4.0 * B * heads * N * N * d
        for tag, fn in (("fwd (no P)", lambda: ops.attn_core_fwd(qkv, o, None, heads, d, N, sc)), ("fwd (+P)", lambda: ops.attn_core_fwd(qkv, o, P, heads, d, N, sc)),
                        ("bwd", lambda: ops.attn_core_bwd(qkv, P, o, do, dS, dqkv, heads, d, N, sc))):
            ms = timeit(fn, n=30)
            print(f"attn_core heads={heads} d={d:3d} {tag:11s}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF")
if "conv3" in what or "conv3w" in what:
    from villandiffusion_amd.lib import B_CONV3, B_CONV3_T
    # (cin, cout, H): the 3x3 convolutions of the ResNet blocks (forward, and the flipped-tap input gradient with the roles of cin / cout swapped)
    if "conv3w" in what:       # BASELINE configs #4 / #5 at per-GPU batch 8: the wide levels
        B = 8
    for cin, cout, H in [(128, 128, 256), (128, 128, 128), (256, 128, 128), (256, 256, 64), (512, 256, 64), (224, 224, 64), (448, 224, 64), (448, 448, 32)] if "conv3w" in what else [(128, 3, 32), (3, 128, 32), (128, 128, 32), (256, 128, 32), (384, 128, 32), (256, 256, 16), (512, 256, 16), (384, 256, 16), (256, 256, 8), (512, 256, 8),
                         (256, 256, 4), (512, 256, 4)]:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin * 9, device="cuda") / math.sqrt(cin * 9)
        out = torch.empty(B, cout, H, H, device="cuda")
        pk = ops.conv3_pack_weights(w, cout, cin) if (cin % 16 == 0 and cout >= 64) else None      # conv_in / conv_out: exact-f32 kernels
        ms = timeit(lambda: ops.conv3x3(x, w, None, out, mode=B_CONV3, a_packed=pk), n=30)
        fl = 2.0 * cout * cin * 9 * B * H * H
        line = f"conv3x3 {cin:4d}->{cout:4d} @{H:2d}^2: fwd {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF"
        if cout % 16 == 0 and pk is not None:
            pkt = ops.conv3_pack_weights(w, cin, cout, transposed=True)
            dx = torch.empty_like(x)
            wt = torch.empty(cin, cout * 9, device="cuda")                 # shape carrier: the kernel reads the packed operand
            ms = timeit(lambda: ops.conv3x3(out, wt, None, dx, mode=B_CONV3_T, a_packed=pkt), n=30)
            line += f" | dgrad {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF"
        print(line)
