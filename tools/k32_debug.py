"""Where does the k32 convolution differ from torch?  Prints error maps by output channel block, image row, image column, image."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3

B, Cin, Cout, H = int(os.environ.get("KB", 64)), int(os.environ.get("KC", 128)), 128, int(os.environ.get("KH", 32))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, H, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
only = os.environ.get("KONLY")
if only is not None:                     # keep ONE input channel: which k-group goes wrong?
    keep = torch.zeros(Cin)
    keep[int(only)] = 1
    x = x * keep[None, :, None, None]
y = F.conv2d(x, w, None, padding=1)
wd = w.cuda().view(Cout, -1)
pk = ops.conv3_pack_weights(wd, Cout, Cin)
out = torch.empty(B, Cout, H, H, device="cuda")
ops.conv3x3(x.cuda(), wd, None, out, a_packed=pk)
torch.cuda.synchronize()
e = (out.cpu() - y).abs()
print("max err", float(e.max()), "ref max", float(y.abs().max()))
print("by image  :", [round(float(v), 3) for v in e.amax(dim=(1, 2, 3))[:8]])
print("by m block:", [round(float(v), 3) for v in e.amax(dim=(0, 2, 3)).view(-1, 16).amax(1)])
print("by row    :", [round(float(v), 3) for v in e.amax(dim=(0, 1, 3))])
print("by col    :", [round(float(v), 3) for v in e.amax(dim=(0, 1, 2))])
