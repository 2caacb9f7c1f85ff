"""Where does the host wait inside a steady-state training step?  Average host time of each call of bench.py's loop body (no syncs added)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from villandiffusion_amd import ops
from villandiffusion_amd.dataset import DatasetLoader
from villandiffusion_amd.loss import LossFn
from villandiffusion_amd.model import DDPM_32_ARCH
from villandiffusion_amd.schedulers import DDPMScheduler
from villandiffusion_amd.trainer import Trainer, shard_indices
from villandiffusion_amd.unet import UNet2DModel
dev = torch.device("cuda", 0)
B = 128
net = UNet2DModel(in_channels=3, out_channels=3, sample_size=32, **DDPM_32_ARCH)
net.reset_parameters(seed=0)
sched = DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, clip_sample=False)
loss_fn = LossFn(sched, "SDE-VP", psi=1, solver_type="sde")
dsl = DatasetLoader("SYNTHETIC-CIFAR10", root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), batch_size=B, seed=0)
dsl.set_poison("BOX_14", "HAT", poison_rate=0.1).prepare_dataset(mode="FIXED")
trainer = Trainer(net, loss_fn, lr=2e-4, total_steps=10000, warmup_steps=500, grad_accum=1)
ids = shard_indices(len(dsl), 0, 0, 1, seed=0)
tgen = torch.Generator(device=dev).manual_seed(100)
acc = {}
def timed(name, fn):
    h = time.perf_counter(); r = fn(); acc.setdefault(name, []).append(time.perf_counter() - h); return r
# instrument trainer internals
orig_p = loss_fn.p_loss_by_keys
loss_fn.p_loss_by_keys = lambda *a, **k: timed("  p_loss (forward launches)", lambda: orig_p(*a, **k))
orig_step = trainer.opt.step
trainer.opt.step = lambda *a, **k: timed("  opt.step", lambda: orig_step(*a, **k))
orig_zero = net.zero_grad
net.zero_grad = lambda *a, **k: timed("  zero_grad", lambda: orig_zero(*a, **k))
orig_bwd = net._run_backward
net._run_backward = lambda *a, **k: timed("  _run_backward", lambda: orig_bwd(*a, **k))
for i in range(40):
    if i == 10:
        torch.cuda.synchronize(); acc.clear(); t0 = time.perf_counter()
    s = (i * B) % (len(ids) - B)
    batch = timed("make_batch", lambda: dsl.make_batch(ids[s:s + B], full=False))
    t = timed("randint", lambda: torch.randint(0, 1000, (B,), device=dev, generator=tgen))
    loss = timed("train_step", lambda: trainer.train_step(batch, t))
h_end = time.perf_counter()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"30 steps: host loop {1e3*(h_end-t0)/30:.2f} ms/step, wall {1e3*(t1-t0)/30:.2f} ms/step")
for k, v in acc.items():
    v = sorted(v)
    print(f"{k:32s} mean {1e3*sum(v)/len(v):7.3f} ms  median {1e3*v[len(v)//2]:7.3f}  max {1e3*v[-1]:7.3f}")
