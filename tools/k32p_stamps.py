"""Where a tile of the persistent 16x16x32 convolution spends its time: in-kernel s_memrealtime / s_memtime stamps (diagnostic build,
-DVD_K32P_STAMPS; see tools/attic/r04_stamps.sh) per workgroup: start, prologue done, then per tile: channel loop done, epilogue issued.
    python tools/k32p_stamps.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T  # noqa: E402

B = 128
stamps = torch.zeros(256 * 32, dtype=torch.int64, device="cuda")
for cin, cout, H, mode in [(128, 128, 32, B_CONV3), (384, 128, 32, B_CONV3), (128, 128, 32, B_CONV3_T), (256, 256, 16, B_CONV3), (512, 256, 16, B_CONV3)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin * 9, device="cuda") / math.sqrt(cin * 9)
    out = torch.empty(B, cout, H, H, device="cuda")
    pk = ops.conv3_pack_weights(w, cout, cin, transposed=(mode == B_CONV3_T))
    wt = w if mode == B_CONV3 else torch.empty(cout, cin * 9, device="cuda")
    for _ in range(20):                                     # steady state (clock, caches)
        ops.conv3x3(x, wt, None, out, mode=mode, a_packed=pk)
    torch.cuda.synchronize()
    ops.FORCE_WS = stamps
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv3x3(x, wt, None, out, mode=mode, a_packed=pk)
    e1.record()
    torch.cuda.synchronize()
    ops.FORCE_WS = None
    assert ops.LAST_GEMM_TILE == 18
    st = stamps.cpu().numpy().reshape(256, 32)
    rt, cy = st[:, :16].astype(np.float64) / 100.0, st[:, 16:].astype(np.float64)      # us (100 MHz), shader cycles
    n = int((st[0, :16] != 0).sum())
    t0 = rt[:, 0].min()
    seg = ["start", "prologue"] + [f"loop{k}" if i == 0 else f"epi{k}" for k in range((n - 2) // 2) for i in (0, 1)]
    print(f"== {cin}->{cout} @{H} mode {mode}: event {e0.elapsed_time(e1) * 1e3:.1f} us; first start .. last end {rt[:, n - 1].max() - t0:.1f} us; "
          f"start skew {rt[:, 0].max() - t0:.1f} us")
    for k in range(1, n):
        d_us = rt[:, k] - rt[:, k - 1]
        d_cy = cy[:, k] - cy[:, k - 1]
        clk = np.median(d_cy / np.maximum(d_us, 1e-3)) / 1e3
        print(f"   {seg[k]:9s} median {np.median(d_us):7.2f} us  p10 {np.percentile(d_us, 10):7.2f}  p90 {np.percentile(d_us, 90):7.2f}   at {np.median(rt[:, k]) - t0:7.2f} us   clock {clk:.2f} GHz")
