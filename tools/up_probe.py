"""A/B of the upsample-fused 3x3 convolution at 32x32 outputs (config #2: 256 -> 256 channels, B = 128): k32 kernel (VD_BX3_K32_UP32=1) vs the 128 x 512 tile."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_bench_util import timeit
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3_UP
B = 128
for cin, cout, H in [(256, 256, 16), (256, 256, 8)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin * 9, device="cuda") / math.sqrt(cin * 9)
    out = torch.empty(B, cout, 2 * H, 2 * H, device="cuda")
    pk = ops.conv3_pack_weights(w, cout, cin)
    ms = timeit(lambda: ops.conv3x3(x, w, None, out, mode=B_CONV3_UP, a_packed=pk), n=30)
    fl = 2.0 * cout * cin * 9 * B * 4 * H * H
    print(f"up-conv {cin}->{cout} @{H}->{2*H}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF")
