"""Persistent 3x3 convolution (conv3_k32p_kernel<..., PS = true>) on the training step's shapes, pre-split input, B = 128: average launch time over
back-to-back launches, and -- with a diagnostic library (tools/build_k32p_diag.sh) -- timing-only ablations / scheduling variants (VD_K32P_FLAGS) and
in-kernel stamps (tile level + per-wave segment sums of the stage pipeline).

    python tools/k32p_probe.py                       # release library: times only
    K32P_LIB=tools/diag/libvillan_hip_k32p_var.so VD_K32P_FLAGS=32 python tools/k32p_probe.py
    K32P_LIB=tools/diag/libvillan_hip_k32p_stamps.so python tools/k32p_probe.py --stamps
VD_K32P_FLAGS bits (diagnostic builds): 2 no patch pipeline, 4 no weight DMA, 8 no epilogue, 16 no MFMAs (2-16: WRONG results, timing only),
32 weight DMA pieces spread over the taps (valid)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from villandiffusion_amd import lib as L  # noqa: E402

if os.environ.get("K32P_LIB"):
    L.LIB_PATH = os.path.join(ROOT, os.environ["K32P_LIB"])
from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T  # noqa: E402

B = int(os.environ.get("K32P_B", "128"))
STAMPS = "--stamps" in sys.argv
CHECK = "--check" in sys.argv
F32IN = "--f32" in sys.argv                 # f32 NCHW input (the converting loader) instead of the pre-split image; modes 0, 1 and 3 (GroupNorm folded)
import hashlib  # noqa: E402
torch.manual_seed(0)
SHAPES = [(128, 128, 32), (256, 128, 32), (384, 128, 32), (256, 256, 16), (512, 256, 16), (384, 256, 16)]
flags = int(os.environ.get("VD_K32P_FLAGS", "0"))
print(f"# lib {os.path.basename(L.LIB_PATH)}  VD_K32P_FLAGS={flags}  B={B}")
stamps = torch.zeros(256 * 32 + 256 * 8 * 8, dtype=torch.int64, device="cuda")
SEG = ["issue", "mfma", "switch", "vmwait", "barrier", "epilogue"]
tot_us = 0.0
for cin, cout, H in SHAPES:
    for mode in ((B_CONV3, B_CONV3_T, "gn") if F32IN else (B_CONV3, B_CONV3_T)):
        gn = mode == "gn"
        if gn:
            mode = B_CONV3
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin * 9, device="cuda") / math.sqrt(cin * 9)
        if os.environ.get("K32P_ZERO"):                     # DVFS check: all-zero activations (1) / also all-zero weights (2)
            x.zero_()
            if os.environ["K32P_ZERO"] == "2":
                w.zero_()
        out = torch.empty(B, cout, H, H, device="cuda")
        pk = ops.conv3_pack_weights(w, cout, cin, transposed=False)
        xp = x if F32IN else ops.presplit_pack(x)
        ss = None
        if gn:
            ss = torch.empty(B, cin, 2, device="cuda")
            ops.groupnorm_stats(x, torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, ss, torch.empty(B * 32, device="cuda"),
                                torch.empty(B * 32, device="cuda"), 32, 1e-6)
        _conv = ops.conv3x3
        ops_conv3x3 = (lambda a_, w_, b_, o_, mode, a_packed: _conv(a_, w_, b_, o_, mode=mode, a_packed=a_packed, gn_ss=ss))
        for _ in range(10):
            ops_conv3x3(xp, w, None, out, mode=mode, a_packed=pk)
        assert ops.LAST_GEMM_TILE == 18
        torch.cuda.synchronize()
        n = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops_conv3x3(xp, w, None, out, mode=mode, a_packed=pk)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tot_us += us
        gf = 2.0 * cout * cin * 9 * B * H * H / 1e9
        line = f"{cin:4d}->{cout:3d} @{H:2d} mode {'2+gn' if gn else mode}: {us:7.1f} us  {3e3 * gf / us:6.0f} TF/s executed  sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]}"
        if CHECK and not F32IN:
            ref = torch.empty_like(out)
            ops.conv3x3(x, w, None, ref, mode=mode, a_packed=pk)
            torch.cuda.synchronize()
            line += f"  bits {'same' if torch.equal(ref, out) else 'DIFFER %.3g' % float((ref - out).abs().max())}"
        print(line, flush=True)
        if STAMPS:
            ops.FORCE_WS = stamps
            stamps.zero_()
            ops.conv3x3(xp, w, None, out, mode=mode, a_packed=pk)
            torch.cuda.synchronize()
            ops.FORCE_WS = None
            st = stamps.cpu().numpy()
            tl = st[:256 * 32].reshape(256, 32)
            rt, cy = tl[:, :16].astype(np.float64) / 100.0, tl[:, 16:].astype(np.float64)
            ns = int((tl[0, :16] != 0).sum())
            t0 = rt[:, 0].min()
            names = ["start", "prologue"] + [f"loop{k}" if i == 0 else f"epi{k}" for k in range((ns - 2) // 2) for i in (0, 1)]
            print(f"     first start .. last end {rt[:, ns - 1].max() - t0:.1f} us; start skew {rt[:, 0].max() - t0:.1f} us")
            for k in range(1, ns):
                d_us, d_cy = rt[:, k] - rt[:, k - 1], cy[:, k] - cy[:, k - 1]
                clk = np.median(d_cy / np.maximum(d_us, 1e-3)) / 1e3
                print(f"     {names[k]:9s} median {np.median(d_us):7.2f} us  p10 {np.percentile(d_us, 10):7.2f}  p90 {np.percentile(d_us, 90):7.2f}   clock {clk:.2f} GHz")
            fs = st[256 * 32:].reshape(256, 8, 8)[:, :, :6].astype(np.float64)          # [workgroup][wave][segment] shader cycles
            tiles = (ns - 2) // 2
            stages = tiles * (cin // 32) * 3
            for grp, nm in ((slice(0, 4), "waves 0-3"), (slice(4, 8), "waves 4-7")):
                m = np.median(fs[:, grp, :].reshape(-1, 6), axis=0) / stages
                print(f"     {nm}: cycles per stage " + "  ".join(f"{SEG[k]} {m[k]:7.0f}" for k in range(6)) + f"   sum {m.sum():7.0f}  (MFMA need 4608 per SIMD = 2 waves)")
print(f"# sum of averages {tot_us:.1f} us")
