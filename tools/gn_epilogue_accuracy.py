"""No-grad forward at batch 128: GroupNorm statistics from the conv epilogue vs the statistics pass, each against the float64 CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.unet_ref import UNet2DModelRef
from villandiffusion_amd.unet import UNet2DModel
torch.manual_seed(0)
ref = UNet2DModelRef()
with torch.no_grad():
    for n, p in ref.named_parameters():
        if "norm" in n:
            p.add_(0.1 * torch.randn_like(p))
net = UNet2DModel()
net.load_state_dict(ref.state_dict())
net.conv_math = "bf16x3"
x = torch.randn(128, 3, 32, 32, generator=torch.Generator().manual_seed(21))
t = torch.randint(0, 1000, (128,), generator=torch.Generator().manual_seed(22))
ref64 = ref.double()
with torch.no_grad():
    y64 = ref64(x.double(), t)[0]
    outs = {}
    for flag in (True, False):
        net.gn_stats_in_epilogue = flag
        outs[flag] = net(x.cuda(), t.cuda())[0].double().cpu()
sc = float(y64.abs().max())
for flag in (True, False):
    print(f"stats in epilogue={flag}: max abs err vs f64 oracle / max|y| = {float((outs[flag] - y64).abs().max()) / sc:.3e}")
print(f"between the two: {float((outs[True] - outs[False]).abs().max()) / sc:.3e}")
