"""Per-shape timing of the implicit-GEMM kernels (HIP events), with timing-only ablation flags.
   python tools/gemm_bench.py"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T, B_PLAIN

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = 128
shapes = [(128, 128, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (256, 256, 4), (256, 128, 32)]
for cin, cout, H in shapes:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin * 9, device="cuda") / math.sqrt(cin * 9)
    bias = torch.randn(cout, device="cuda")
    out = torch.empty(B, cout, H, H, device="cuda")
    flops = 2.0 * cout * cin * 9 * B * H * H
    line = f"conv3x3 {cin:4d}->{cout:4d} @{H:2d}^2: "
    ms = timeit(lambda: ops.conv3x3(x, w, bias, out, mode=B_CONV3, tile=0))
    print(line + f"AUTO (patch kernel where eligible): {flops / ms / 1e9:6.1f} TF ({ms*1e3:.0f} us)")
    for dbg, nm in ((17, "patch noload"), (25, "patch noload+nostore")):
        ms = timeit(lambda: ops.conv3x3(x, w, bias, out, mode=B_CONV3, tile=0, debug=dbg))
        print(line + f"{nm}: {flops / ms / 1e9:6.1f} TF")
    ms = timeit(lambda: ops.conv3x3(x, w, bias, out, mode=B_CONV3_T, tile=0))
    print(line + f"AUTO dgrad-style (flipped taps): {flops / ms / 1e9:6.1f} TF")
    for tile in (1,):
        res = []
        for dbg in (0, 1, 2, 3, 4, 9):
            ms = timeit(lambda: ops.conv3x3(x, w, bias, out, mode=B_CONV3, tile=tile, debug=dbg))
            res.append(f"{flops / ms / 1e9:6.1f}")
        print(line + f"tile{tile} TF [full, noload, noepi, noload+noepi, nomfma, noload+nostore]: " + " ".join(res))
    dy = torch.randn(B, cout, H, H, device="cuda")
    dw = torch.empty(cout, cin * 9, device="cuda")
    ws = torch.empty(max(ops.wgrad_ws_floats(cout, cin, 9, B, H * H), 4), device="cuda")
    for splits in (0,):
        try:
            need = 1
            ms = timeit(lambda: ops.conv_wgrad(dy, x, dw, B_CONV3, torch.empty(splits * cout * cin * 9 + 4, device="cuda") if splits > 1 else ws, splits=splits))
            print(f"   wgrad splits={splits:2d}: {flops / ms / 1e9:6.1f} TF  ({ms*1e3:.0f} us)")
        except Exception as e:
            print("   wgrad splits", splits, "failed", str(e)[:80])

print("---- 1x1 wgrad (PLAIN) ----")
for cin, cout, H in [(256, 256, 16), (256, 768, 16), (512, 256, 16), (384, 128, 32), (256, 256, 8)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    dy = torch.randn(B, cout, H, H, device="cuda")
    dw = torch.empty(cout, cin, device="cuda")
    flops = 2.0 * cout * cin * B * H * H
    for tile in (1, 3):
        for splits in (0, 8, 32):
            ws = torch.empty(max(64 * cout * cin, 4), device="cuda")
            try:
                ms = timeit(lambda: ops.conv_wgrad(dy, x, dw, B_PLAIN, ws, splits=splits, tile=tile))
                print(f"1x1 wgrad {cin}->{cout}@{H}: tile{tile} splits={splits}: {flops / ms / 1e9:6.1f} TF ({ms*1e3:.0f} us)")
            except Exception as e:
                print("fail", str(e)[:60])
