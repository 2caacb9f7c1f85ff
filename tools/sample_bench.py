"""DDIM-N sampling throughput of the CIFAR10 UNet at batch 128, with / without the folded GroupNorm loaders.
   python tools/sample_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops
from villandiffusion_amd.pipelines import DDIMPipeline
from villandiffusion_amd.schedulers import DDIMScheduler
from villandiffusion_amd.unet import UNet2DModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
net = UNet2DModel()
net.reset_parameters(0)
init = torch.empty(128, 3, 32, 32, device="cuda")
ops.randn(init, 7, 0)
for math in (("bf16x3",) if len(sys.argv) > 2 else ("bf16x3", "f32")):
    for fuse in (True,) if len(sys.argv) > 2 else (False, True):
        net.conv_math, net.fuse_gn_inference = math, fuse
        pipe = DDIMPipeline(net, DDIMScheduler(clip_sample=False))
        pipe(batch_size=128, init=init, num_inference_steps=5, return_tensor=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = pipe(batch_size=128, init=init, num_inference_steps=n, return_tensor=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{math} fuse_gn={fuse}: {1e3 * dt / n:.3f} ms/step ({128 * 1000 / n / dt / 1000 * n:.1f} img/s at {n} steps); checksum {float(o.double().sum()):.6f}", flush=True)
