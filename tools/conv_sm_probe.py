"""The 8x8 / 4x4 3x3 convolutions of config #2 at B = 128 (forward and input gradient; 256 -> 256 and 512 -> 256): average launch time over back-to-back
launches incl. whatever epilogue launch the path needs.  VD_CONV_SM_OFF=1: the split-K kernels of rounds 2-5 (conv3_bx3_kernel + splitk_epilogue4);
default: conv3_sm_kernel (whole K per workgroup).    python tools/conv_sm_probe.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from villandiffusion_amd import lib as _L  # noqa: E402

if os.environ.get("SM_LIB"):                        # a diagnostic library (tools/build_k32p_diag.sh): SM_LIB=tools/diag/libvillan_hip_sm_stamps.so ... --stamps
    _L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["SM_LIB"])
from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T  # noqa: E402

B = 128
STAMPS = "--stamps" in sys.argv
SEG = ["issue", "convert", "mfma", "switch", "vmwait", "barrier", "epilogue", "total"]
stamps = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
torch.manual_seed(0)
tot = 0.0
for cin, cout, S in [(256, 256, 8), (512, 256, 8), (256, 256, 4), (512, 256, 4)]:
    for mode in (B_CONV3, B_CONV3_T):
        ci, co = (cin, cout) if mode == B_CONV3 else (cout, cin)          # the input gradient of cin -> cout contracts cout channels
        x = torch.randn(B, ci, S, S, device="cuda")
        w = torch.randn(co, ci * 9, device="cuda") / math.sqrt(ci * 9)
        bias = torch.randn(co, device="cuda") if mode == B_CONV3 else None
        res = torch.randn(B, co, S, S, device="cuda") if mode == B_CONV3 else None
        out = torch.empty(B, co, S, S, device="cuda")
        pk = ops.conv3_pack_weights(w, co, ci)
        for _ in range(10):
            ops.conv3x3(x, w, bias, out, mode=mode, residual=res, a_packed=pk)
        tile = ops.LAST_GEMM_TILE
        torch.cuda.synchronize()
        n = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.conv3x3(x, w, bias, out, mode=mode, residual=res, a_packed=pk)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tot += us
        gf = 2.0 * co * ci * 9 * B * S * S / 1e9
        print(f"{ci:4d}->{co:3d} @{S} mode {mode} tile {tile:2d}: {us:7.1f} us  {3e3 * gf / us:6.0f} TF/s executed  checksum {float(out.double().sum()):.6e}", flush=True)
        if STAMPS and tile == 20:
            ops.FORCE_WS = stamps
            stamps.zero_()
            ops.conv3x3(x, w, bias, out, mode=mode, residual=res, a_packed=pk)
            torch.cuda.synchronize()
            ops.FORCE_WS = None
            fs = stamps.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
            fs = fs[fs[:, 0, 7] > 0]                                       # workgroups that ran
            stages = 3 * (ci // 32)
            m = np.median(fs.reshape(-1, 8), axis=0)
            print(f"     {fs.shape[0]} workgroups, {stages} stages; cycles per stage: " + "  ".join(f"{SEG[k]} {m[k] / stages:6.0f}" for k in range(6))
                  + f"   epilogue {m[6]:6.0f}  loop total {m[7] - m[6]:7.0f} = {(m[7] - m[6]) / stages:6.0f} per stage")
print(f"# sum {tot:.1f} us  (VD_CONV_SM_OFF={os.environ.get('VD_CONV_SM_OFF', '0')})")
