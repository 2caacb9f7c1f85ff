"""The 8x8 / 4x4 3x3 convolutions of config #2 at B = 128 (forward and input gradient; 256 -> 256 and 512 -> 256): average launch time over back-to-back
launches incl. whatever epilogue launch the path needs.  VD_CONV_SM_OFF=1: the split-K kernels of rounds 2-5 (conv3_bx3_kernel + splitk_epilogue4);
default: conv3_sm_kernel (whole K per workgroup).    python tools/conv_sm_probe.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T  # noqa: E402

B = 128
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
print(f"# sum {tot:.1f} us  (VD_CONV_SM_OFF={os.environ.get('VD_CONV_SM_OFF', '0')})")
