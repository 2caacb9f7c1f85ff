"""Debug: UNet fwd+bwd with pre-split operands on / off at several batch sizes; reports non-finite gradients per parameter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops
from villandiffusion_amd.unet import UNet2DModel

net = UNet2DModel()
net.reset_parameters(seed=3)
for B in (int(a) for a in (sys.argv[1:] or ["64", "96", "128", "48"])):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, 32, 32, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(2)).cuda()
    dy = (torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(3)) * 1e-4).cuda()
    res = {}
    for ps in (False, True, True):
        net.presplit = ps
        net.zero_grad()
        ops.profile_start()
        y = net(x, t, return_dict=False)[0]
        y.backward(dy)
        torch.cuda.synchronize()
        names = sorted({r["name"] for r in ops.profile_stop() if "_ps_" in r["name"] or r["name"].endswith("true>")})
        gr = net.flat_grad.detach().clone()
        bad = [n for n, p in net.named_parameters() if not bool(torch.isfinite(p.grad).all())]
        res[ps] = gr
        print(f"B={B} presplit={ps}: finite={bool(torch.isfinite(gr).all())} bad params: {bad[:6]}{'...' if len(bad) > 6 else ''} ({len(bad)}); ps kernels: {len(names)}")
    if bool(torch.isfinite(res[True]).all()):
        print(f"   |ps - off| / |off| = {float((res[True] - res[False]).norm() / res[False].norm()):.2e}")
