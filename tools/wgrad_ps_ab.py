"""Same-box A/B of the grouped 3x3 weight gradients of config #2 at B = 128: the converting 16x16x32 kernel (wgrad_k32_group_kernel, f32 operands) against
the pre-split kernel (wgrad_ps_group_kernel: LDS-DMA + ds_read_b64_tr_b16), interleaved rounds; plus the conversion pass (presplit_pack) an operand
costs when its producer cannot write the image, and the GroupNorm producers (f32 output vs pre-split output).
   python tools/wgrad_ps_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3

DEV = torch.device("cuda")
B = 128
# (side, [(Cin, Cout), ...]): the plain 3x3 layers of the DDPM-CIFAR10-32 UNet (block_out_channels (128, 256, 256, 256), 2 layers per block)
GROUPS = {
    32: [(128, 128)] * 7 + [(384, 128), (256, 128), (256, 128)] + [(128, 128)] * 3,          # conv1 / conv2 of the 32x32 resnets (down 0, up 3)
    16: [(128, 256), (256, 256), (256, 256), (256, 256)] + [(512, 256), (256, 256)] * 2 + [(384, 256), (256, 256)],
}


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for S, layers in GROUPS.items():
    df, dp, keep = [], [], []
    flops = 0.0
    for cin, cout in layers:
        x = torch.randn(B, cin, S, S, device=DEV)
        dy = torch.randn(B, cout, S, S, device=DEV)
        dwf, dwp = torch.zeros(cout, cin * 9, device=DEV), torch.zeros(cout, cin * 9, device=DEV)
        xp, dyp = ops.presplit_pack(x), ops.presplit_pack(dy)
        df.append(ops.wgrad_desc(dy, x, dwf, B_CONV3, None, accumulate=True, math_mode=1))
        dp.append(ops.wgrad_desc(dyp, xp, dwp, B_CONV3, None, accumulate=True, math_mode=1))
        keep.append((x, dy, xp, dyp, dwf, dwp))
        flops += 2.0 * cout * cin * 9 * B * S * S
    res = {"k32": [], "ps": []}
    for rnd in range(3):
        res["k32"].append(timed(lambda: ops.conv_wgrad_group(df, DEV)))
        res["ps"].append(timed(lambda: ops.conv_wgrad_group(dp, DEV)))
    same = all(torch.equal(k[4], k[5]) for k in keep)
    for name in ("k32", "ps"):
        ms = min(res[name])
        print(f"{S}x{S} grouped 3x3 wgrad, {len(layers)} layers, {flops / 1e9:.0f} GFLOP: {name:4s} " + " ".join(f"{t:.3f}" for t in res[name]) +
              f" ms  -> {flops / ms / 1e9:.0f} TF/s algorithmic ({3 * flops / ms / 1e12:.2f} PF/s executed)")
    print(f"    bit-identical results: {same}")
    x, dy = keep[0][0], keep[0][1]
    t_pack = timed(lambda: ops.presplit_pack(dy, out=keep[0][3]))
    print(f"    presplit_pack of one [{B}, {dy.shape[1]}, {S}, {S}] tensor: {t_pack * 1e3:.1f} us ({8.0 * dy.numel() / t_pack / 1e9:.0f} GB/s)")

G = 32
for C, S in ((128, 32), (256, 32), (384, 32), (256, 16), (512, 16), (384, 16)):
    x = torch.randn(B, C, S, S, device=DEV)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    y = torch.empty_like(x)
    yp = ops.presplit_empty(x.shape, DEV)
    m, r = torch.empty(B * G, device=DEV), torch.empty(B * G, device=DEV)
    a, b = [], []
    for rnd in range(3):
        a.append(timed(lambda: ops.groupnorm_fwd(x, gamma, beta, y, m, r, G, 1e-6, True), 10))
        b.append(timed(lambda: ops.groupnorm_fwd_presplit(x, gamma, beta, yp, m, r, G, 1e-6, True), 10))
    byt = 8.0 * x.numel()
    print(f"GroupNorm+SiLU fwd C={C} {S}x{S}: f32 out {min(a) * 1e3:.1f} us ({byt / min(a) / 1e9:.0f} GB/s) | pre-split out {min(b) * 1e3:.1f} us ({byt / min(b) / 1e9:.0f} GB/s)")
