"""Experiment: one optimiser step's forward + backward at B = 128 as ONE pass vs as N concurrent sub-batches (each a HIP-graph replay of the micro-step on
its own stream, its own network object with the same weights): do complementary kernels of different sub-batches (MFMA-bound convolutions beside
HBM-bound GroupNorm / 1x1 passes) overlap enough to beat the single pass?   python tools/halves_probe.py [parts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd.loss import LossFn
from villandiffusion_amd.schedulers import DDPMScheduler
from villandiffusion_amd.trainer import Trainer, GraphedMicroStep
from villandiffusion_amd.unet import UNet2DModel

dev = torch.device("cuda")
B = 128
g = torch.Generator().manual_seed(0)
x0 = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
R = torch.zeros(B, 3, 32, 32, device=dev)
noise = torch.randn(B, 3, 32, 32, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)


def make(parts):
    nets, graphs, streams = [], [], []
    n = B // parts
    for p in range(parts):
        net = UNet2DModel()
        net.reset_parameters(seed=0)
        lf = LossFn(DDPMScheduler(), "SDE-VP", psi=1)
        tr = Trainer(net, lf, lr=1e-4, total_steps=100, warmup_steps=0, grad_accum=1 << 20, graph_micro_step=True)
        sl = slice(p * n, (p + 1) * n)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            gm = GraphedMicroStep(tr, x0[sl], R[sl], noise[sl], t[sl])
        torch.cuda.synchronize()
        nets.append((net, tr))
        graphs.append((gm, sl))
        streams.append(s)
    return nets, graphs, streams


def run(parts, iters=20):
    nets, graphs, streams = make(parts)
    main = torch.cuda.current_stream(dev)

    def once():
        for (gm, sl), s in zip(graphs, streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                gm.graph.replay()
        for s in streams:
            main.wait_stream(s)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{parts} concurrent sub-batch(es) of {B // parts}: {ms:.3f} ms per forward + backward of {B} images (HIP-graph replays)", flush=True)
    del nets, graphs
    torch.cuda.empty_cache()
    return ms


for parts in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "1", "2"])]:
    run(parts)
