"""Where a K-step of the split-precision 3x3 weight gradient (wgrad_bx3_body) spends its time: per-wave s_memtime sums around the loop's
segments (diagnostic build -DVD_WG_STAMPS, tools/diag/libvillan_hip_wgstamps.so; see tools/attic/r04_wg_stamps.sh).  The grouped launch of
BASELINE config #2's ten 32x32 layers (or the four 16x16 ones) at B = 128.
    python tools/wg_stamps.py [32|16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from villandiffusion_amd import lib as L  # noqa: E402
from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = 128
layers = [(128, 128)] * 7 + [(256, 128)] * 3 if H == 32 else [(256, 256)] * 7 + [(512, 256)] * 3
dev = torch.device("cuda")
keep, descs = [], []
for cin, cout in layers:
    dy = torch.randn(B, cout, H, H, device=dev)
    x = torch.randn(B, cin, H, H, device=dev)
    dw = torch.zeros(cout, cin * 9, device=dev)
    keep.append((dy, x, dw))
    d = ops.wgrad_desc(dy, x, dw, B_CONV3, None, accumulate=True, math_mode=1)
    assert ops.wgrad_group_class(d)
    descs.append(d)
for _ in range(10):
    ops.conv_wgrad_group(descs, dev)
torch.cuda.synchronize()
lib = L.load()
lib.vd_wg_stamps_set.restype, lib.vd_wg_stamps_set.argtypes = C.c_int, [C.c_void_p]
NB = 1 << 15
stamps = torch.zeros(NB * 4 * 12, dtype=torch.int64, device=dev)
assert lib.vd_wg_stamps_set(stamps.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv_wgrad_group(descs, dev)
e1.record()
torch.cuda.synchronize()
lib.vd_wg_stamps_set(None)
st = stamps.cpu().numpy().reshape(NB, 4, 12).astype(np.float64)
used = st[:, 0, 0] != 0
n = int(used.sum())
st = st[used]
rt0, rt1 = st[:, :, 0] / 100.0, st[:, :, 1] / 100.0                  # us
t_first = rt0.min()
print(f"== {H}x{H}: {len(layers)} layers, event {e0.elapsed_time(e1) * 1e3:.1f} us (compute + slab reduce); {n} workgroups recorded; "
      f"first start .. last end {rt1.max() - t_first:.1f} us")
dur = (rt1 - rt0)[:, 0]
print(f"   workgroup duration us: median {np.median(dur):.1f} p10 {np.percentile(dur, 10):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; "
      f"start time p50 {np.median(rt0[:, 0]) - t_first:.1f} p90 {np.percentile(rt0[:, 0], 90) - t_first:.1f} max {rt0[:, 0].max() - t_first:.1f}")
cyc = st[:, :, 4] - st[:, :, 2]
clk = np.median(cyc[:, 0] / np.maximum(dur, 1e-3)) / 1e3
steps = st[:, :, 10]
print(f"   K-steps per workgroup: median {np.median(steps):.0f} min {steps.min():.0f} max {steps.max():.0f}; shader clock {clk:.2f} GHz")
per = lambda k: np.median(st[:, :, k] / np.maximum(steps, 1))            # noqa: E731
tot = sum(per(k) for k in range(5, 10))
print(f"   cycles per K-step and wave (median): loads-issue {per(5):.0f}  mfma+lds-reads {per(6):.0f}  barrier1 {per(7):.0f}  convert+store {per(8):.0f}  "
      f"barrier2 {per(9):.0f}  = {tot:.0f}  (36 MFMAs of 32 cycles = 1152 per wave; 3 waves per SIMD share one matrix pipe)")
pro = np.median((st[:, :, 3] - st[:, :, 2]) - st[:, :, 5:10].sum(-1))
epi = np.median(st[:, :, 4] - st[:, :, 3])
print(f"   outside the loop (median cycles): prologue {pro:.0f}  epilogue {epi:.0f}  of {np.median(cyc):.0f} total")
# occupancy over time: how many workgroups are alive in each 20 us bin
edges = np.arange(0, rt1.max() - t_first + 20, 20)
alive = [(int(((rt0[:, 0] - t_first < b + 20) & (rt1[:, 0] - t_first > b)).sum())) for b in edges[:-1]]
print("   workgroups alive per 20 us bin:", alive)
