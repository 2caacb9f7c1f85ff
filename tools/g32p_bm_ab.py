"""Per-shape timing of the persistent 1x1 kernel on config #2's 1x1 products at B = 128 (forward and input gradient); run once per setting of
VD_G32P_BM256 (0: 128 x 256 tiles everywhere, 1: 256 x 128 tiles where M % 256 == 0) / VD_G32P_DEPTH (2 | 3 stages of load lead); the hash column is
over the outputs (the settings only change the schedule: it must not move).   python tools/g32p_bm_ab.py"""
import hashlib
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import lib as _L
if os.environ.get("G32P_LIB"):                      # a diagnostic / previous-round library for A/B runs (tools/diag/)
    _L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["G32P_LIB"])
from villandiffusion_amd import ops
from villandiffusion_amd.lib import A_COL, B_PLAIN

DEV = torch.device("cuda")
B = 128
SHAPES = [("qkv", 256, 768, 16), ("shortcut x3 tiles", 128, 256, 32), ("attn out / shortcut", 256, 256, 16), ("shortcut", 128, 256, 16), ("shortcut", 512, 256, 16), ("shortcut", 384, 256, 16),
          ("shortcut", 384, 128, 32), ("shortcut", 256, 128, 32)]


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


tot = 0.0
torch.manual_seed(0)                                  # the hash column is comparable across processes / libraries
for name, cin, cout, S in SHAPES:
    HW = S * S
    x = torch.randn(B, cin, S, S, device=DEV)
    w = torch.randn(cout, cin, device=DEV) / math.sqrt(cin)
    y = torch.empty(B, cout, S, S, device=DEV)
    pk = ops.conv3_pack_weights(w, cout, cin, taps=1)
    pkt = ops.conv3_pack_weights(w, cin, cout, transposed=True, taps=1)
    t_f = timed(lambda: ops.conv1x1(x, w, None, y, a_packed=pk))
    tile_f = ops.LAST_GEMM_TILE
    dx = torch.empty_like(x)
    t_b = timed(lambda: ops.gemm(w, y, dx, M=cin, N=B * HW, K=cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=cin, ldb=HW, b_bstride=cout * HW, ldd=HW,
                                 d_bstride=cin * HW, a_packed=pkt))
    tile_b = ops.LAST_GEMM_TILE
    byt = 4.0 * (x.numel() + y.numel())
    tot += t_f + t_b
    hsh = hashlib.sha1(y.cpu().numpy().tobytes() + dx.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"{hsh} {name:20s} {cin:4d} -> {cout:4d} @{S:2d}: forward {t_f:6.1f} us (tile {tile_f}, {byt / t_f / 1e6:.2f} TB/s)   input gradient {t_b:6.1f} us (tile {tile_b}, {byt / t_b / 1e6:.2f} TB/s)")
print(f"VD_G32P_BM256={os.environ.get('VD_G32P_BM256', '1')} VD_G32P_DEPTH={os.environ.get('VD_G32P_DEPTH', '3')}: sum {tot:.1f} us")
