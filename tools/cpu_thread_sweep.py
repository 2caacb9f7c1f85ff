"""CPU-oracle thread sweep (round-5 review, Weak 9): bench.py's cpu_baseline leg runs the oracle's training step (fwd + bwd + clip + Adam, batch 128) on
<= 32 host threads because more are SLOWER on the GPU box's 256-thread host; this records that claim for the benched batch.  One process per thread
count (torch / oneDNN size their pools at first use); 1 warm-up step at batch 2 + 2 timed steps at batch 128.
    python tools/cpu_thread_sweep.py [16 32 64 128 256]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import torch
n = int(sys.argv[1])
torch.set_num_threads(n)
from oracle.loss_ref import LossFnRef, SDE_VP
from oracle.schedulers_ref import DDPMSchedulerRef
from oracle.unet_ref import UNet2DModelRef
torch.manual_seed(0)
net = UNet2DModelRef()
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
lf = LossFnRef(DDPMSchedulerRef(), SDE_VP, psi=1, solver_type="sde")
def step(b, seed):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(b, 3, 32, 32, generator=g) * 2 - 1
    R = torch.zeros(b, 3, 32, 32)
    t = torch.randint(0, 1000, (b,), generator=g)
    loss = lf.p_loss(net, x0, R, t, noise=torch.randn(x0.shape, generator=g))
    opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0); opt.step()
step(2, 1)
t0 = time.perf_counter()
for i in range(2):
    step(128, 100 + i)
dt = (time.perf_counter() - t0) / 2
print(f"threads {n:4d}: {dt:7.2f} s per B = 128 step  {128 / dt:7.2f} train img/s", flush=True)
""" % ROOT

counts = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 256]
print(f"# host: {len(os.sched_getaffinity(0))} schedulable threads; oracle training step (fwd + bwd + clip + Adam), batch 128, plain torch fp32 (oneDNN)")
for n in counts:
    if n > len(os.sched_getaffinity(0)):
        print(f"threads {n:4d}: skipped (host has fewer)")
        continue
    r = subprocess.run([sys.executable, "-c", CHILD, str(n)], capture_output=True, text=True, timeout=900)
    print(r.stdout.strip() or ("FAILED: " + r.stderr[-300:]), flush=True)
