import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import schedulers as S
from villandiffusion_amd.loss import LossFn
from villandiffusion_amd.trainer import Trainer
from villandiffusion_amd.unet import UNet2DModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
stream = int(sys.argv[2]) if len(sys.argv) > 2 else 1
group = int(sys.argv[3]) if len(sys.argv) > 3 else 1
net = UNet2DModel()
net.reset_parameters(seed=3)
net.wgrad_stream = bool(stream)
net.group_wgrad = bool(group)
tr = Trainer(net, LossFn(S.DDPMScheduler(), "SDE-VP", psi=1), lr=1e-3, total_steps=10, warmup_steps=0, grad_accum=2)
g = torch.Generator().manual_seed(5)
for i in range(4):
    x0 = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).cuda()
    R_ = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).cuda()
    eps = torch.randn(B, 3, 32, 32, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    l = tr.train_step({"target": x0, "pixel_values": R_}, t, noise=eps)
    torch.cuda.synchronize()
    print(f"B={B} stream={stream} group={group} micro-step {i}: loss {float(l):.5f}", flush=True)
print("OK", flush=True)
