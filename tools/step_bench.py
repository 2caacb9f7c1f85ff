"""Time one training step (fwd + bwd + clip + Adam) of a named UNet configuration and list the MFMA kernels by time.
   python tools/step_bench.py [cifar10|celebahq256|ldm64] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops, schedulers as S
from villandiffusion_amd.loss import LossFn
from villandiffusion_amd.trainer import Trainer
from villandiffusion_amd.unet import UNet2DModel

CFG = {
    "cifar10": (dict(), 128),
    "celebahq256": (dict(sample_size=256, block_out_channels=(128, 128, 256, 256, 512, 512),
                         down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                         up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4), 8),
    "ldm64": (dict(sample_size=64, block_out_channels=(224, 448, 672, 896), attention_head_dim=32,
                   down_block_types=("DownBlock2D",) + ("AttnDownBlock2D",) * 3,
                   up_block_types=("AttnUpBlock2D",) * 3 + ("UpBlock2D",)), 8),
}
CFG["ncsnpp32"] = (None, 128)
name = sys.argv[1] if len(sys.argv) > 1 else "celebahq256"
cfg, B = CFG[name]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
if name == "ncsnpp32":        # SDE-VE: NCSN++ (model.py:839-857) with the VE loss
    from villandiffusion_amd.model import NCSNPP_32_ARCH
    from villandiffusion_amd.ncsnpp import NCSNppModel
    net = NCSNppModel(in_channels=3, out_channels=3, sample_size=32, **NCSNPP_32_ARCH)
    net.reset_parameters(0)
    sched = S.ScoreSdeVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    tr = Trainer(net, LossFn(sched, "SDE-VE", psi=0), lr=2e-4, total_steps=1000)
else:
    net = UNet2DModel(**cfg)
    net.reset_parameters(0)
    sched = S.DDPMScheduler()
    tr = Trainer(net, LossFn(sched, "SDE-VP"), lr=2e-4, total_steps=1000)
Sz = net.sample_size
g = torch.Generator(device="cuda").manual_seed(0)
x0 = torch.randn(B, 3, Sz, Sz, device="cuda", generator=g)
R = torch.zeros_like(x0)
t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
if name == "ncsnpp32":
    x0 = x0 * 0.5 + 0.5


def step():
    return tr.train_step({"target": x0, "pixel_values": R}, t)


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{name} B={B}: {ms:.1f} ms/step = {B / ms * 1e3:.1f} img/s; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
ws0, net.wgrad_stream = getattr(net, "wgrad_stream", False), False      # per-launch durations: weight gradients on the launch stream
ops.profile_start()
step()
torch.cuda.synchronize()
rec = ops.profile_stop()
if hasattr(net, "wgrad_stream"):
    net.wgrad_stream = ws0
agg = {}
for r in rec:                                                           # records of ops._timed / ops.gemm: dicts {name, flops, bytes, e0, e1, kind}
    d = agg.setdefault(r["name"], [0, 0.0, 0.0])
    d[0] += 1; d[1] += r["flops"]; d[2] += r["e0"].elapsed_time(r["e1"])
tot = sum(v[2] for v in agg.values())
print(f"profiled kernels: {tot:.1f} ms, {sum(v[1] for v in agg.values()) / tot / 1e9:.1f} TF avg")
for nm, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:int(os.environ.get("STEP_BENCH_TOP", 14))]:
    print(f"  {nm:55s} n={v[0]:3d} {v[2]:8.2f} ms {v[1] / v[2] / 1e9:6.1f} TF")
if os.environ.get("STEP_BENCH_SHAPES"):                               # GEMM-family launches by (kernel, M, K, NP, batch, OH, OW)
    ag2 = {}
    for r in rec:
        if "shape" in r:
            d = ag2.setdefault((r["name"],) + tuple(r["shape"]), [0, 0.0, 0.0])
            d[0] += 1; d[1] += r["flops"]; d[2] += r["e0"].elapsed_time(r["e1"])
    print("by shape (kernel, M, K, NP, batch, OH, OW):")
    for k, v in sorted(ag2.items(), key=lambda kv: -kv[1][2])[:int(os.environ["STEP_BENCH_SHAPES"])]:
        print(f"  {k[0]:45s} {str(k[1:]):40s} n={v[0]:3d} {v[2]:8.3f} ms {v[1] / v[2] / 1e9:6.1f} TF")

# ---- BASELINE config #1 shape: batch 4 with gradient accumulation (a micro-step is launch-bound when launched eagerly) ----
if name == "cifar10" and os.environ.get("STEP_BENCH_MICRO", "1") == "1":
    import time
    for graphed in (False, True):
        tr4 = Trainer(net, LossFn(sched, "SDE-VP"), lr=2e-4, total_steps=1000, grad_accum=32, graph_micro_step=graphed)
        xb, Rb, tb = x0[:4].contiguous(), R[:4].contiguous(), t[:4].contiguous()
        nz = torch.randn_like(xb)
        for _ in range(34):
            tr4.train_step({"target": xb, "pixel_values": Rb}, tb, noise=nz)
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        for _ in range(64):                                             # two optimiser steps of 32 micro-steps
            tr4.train_step({"target": xb, "pixel_values": Rb}, tb, noise=nz)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - h0) / 64
        print(f"config #1 micro-step (B=4, G=32), {'HIP graph replay' if graphed else 'eager launches  '}: {dt * 1e3:.2f} ms = {4 / dt:.0f} img/s")
