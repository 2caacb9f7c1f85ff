"""Throughput of the DROP-IN sampling job (sampling_io.batch_sampling_save, what VillanDiffusion.py --mode measure runs: reference :1062-1067), full-size
CIFAR10 UNet, 1000-step DDPM, chunks of --eval_max_batch = 128:  python tools/measure_path_bench.py [n_images]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import sampling_io as SIO
from villandiffusion_amd.model import DDPM_32_ARCH
from villandiffusion_amd.pipelines import DDPMPipeline
from villandiffusion_amd.schedulers import DDPMScheduler
from villandiffusion_amd.unet import UNet2DModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
net = UNet2DModel(in_channels=3, out_channels=3, sample_size=32, **DDPM_32_ARCH)
net.reset_parameters(seed=0)
init = torch.randn(n, 3, 32, 32, generator=torch.Generator().manual_seed(0))
for host_rng, streams in ((False, "4"), (False, "1"), (True, "1")):
    os.environ["VILLAN_SAMPLER_STREAMS"] = streams
    sch = DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02)
    if not host_rng:
        sch.device_rng_seed = 0                     # what measure() selects (VILLAN_HOST_RNG=1 keeps the CPU generator)
    pipe = DDPMPipeline(net, sch)
    with tempfile.TemporaryDirectory() as d:
        SIO.batch_sampling_save(min(n, 256), pipe, d, init=init[:256], max_batch_n=128, rng=torch.Generator().manual_seed(0), num_inference_steps=3)   # warm-up: graphs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        SIO.batch_sampling_save(n, pipe, d, init=init, max_batch_n=128, rng=torch.Generator().manual_seed(0), num_inference_steps=1000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert len(os.listdir(d)) == n
    print(f"batch_sampling_save, {n} images, DDPM-1000, {'CPU-generator' if host_rng else 'in-kernel'} noise, {streams} chunk(s) at a time: "
          f"{n / dt:.2f} img/s ({dt:.1f} s incl. {n} PNG files)")
