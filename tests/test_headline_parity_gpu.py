"""Parity at the lengths the headline metric is quoted on (BASELINE.json: train img/s + 1000-step DDPM sample img/s), in BOTH
convolution arithmetics (split-precision "bf16x3", the default, and exact "f32"):

* the full 1000-step DDPM reverse process (reference VillanDiffusion.py:843-852 -> DDPMPipeline; CPU-generator noise like
  VillanDiffusion.py:621-624), denoised images within the 1e-3 north_star states, for the variance the `--sched DDPM-SCHED`
  recipe builds (fixed_small, model.py:615) and the one the hub checkpoint ships (fixed_large, model.py:654);
* 20 optimiser steps at batch 8 with the reference's schedule (lr 2e-4, 500 warm-up steps, clip 1.0, Adam; VillanDiffusion.py:1141-1176):
  per-step loss and the L2 error of the accumulated parameter update;
* config #1 as SURVEY §8d writes it: `--batch 4` -> gradient accumulation 32 (VillanDiffusion.py:287), one sync step.

The oracle (oracle/, plain fp32 torch on the host cores) runs ONCE per test and both arithmetics are held to it.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import schedulers_ref as R  # noqa: E402
from oracle.loss_ref import LossFnRef, SDE_VP  # noqa: E402
from oracle.unet_ref import UNet2DModelRef  # noqa: E402
from villandiffusion_amd import schedulers as S  # noqa: E402
from villandiffusion_amd.loss import LossFn  # noqa: E402
from villandiffusion_amd.pipelines import DDPMPipeline  # noqa: E402
from villandiffusion_amd.trainer import Trainer  # noqa: E402
from villandiffusion_amd.unet import UNet2DModel  # noqa: E402

ARITH = ("bf16x3", "f32")


@pytest.fixture(scope="module")
def ref0():
    torch.manual_seed(0)
    return UNet2DModelRef()


def _net_like(ref, conv_math):
    net = UNet2DModel()
    net.load_state_dict(ref.state_dict())
    net.conv_math = conv_math
    return net


@pytest.mark.timeout(900)
@pytest.mark.parametrize("variance_type,clip", [("fixed_small", True), ("fixed_large", False)])
def test_ddpm_1000_steps_match_oracle(ref0, variance_type, clip):
    """1000 UNet evaluations + 1000 scheduler updates with 999 noise draws from the same CPU generator, B = 2."""
    init = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    sref = R.DDPMSchedulerRef(clip_sample=clip, variance_type=variance_type)
    with torch.no_grad():
        x_ref = R.sample_loop(ref0, sref, init.clone(), 1000, generator=torch.Generator().manual_seed(5))
    img_ref = (x_ref / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    assert np.isfinite(img_ref).all() and float(img_ref.std()) > 1e-3          # a non-degenerate image, not a clamped constant
    for conv_math in ARITH:
        net = _net_like(ref0, conv_math)
        sched = S.DDPMScheduler(clip_sample=clip, variance_type=variance_type)
        out = DDPMPipeline(net, sched)(batch_size=2, generator=torch.Generator().manual_seed(5), init=init, num_inference_steps=1000,
                                       output_type=None)
        assert torch.equal(sched.timesteps, sref.timesteps) and sched.timesteps.dtype == torch.int64 and len(sched.timesteps) == 1000
        err = float(np.abs(out.images - img_ref).max() / np.abs(img_ref).max())
        print(f"[parity] DDPM-1000 {variance_type} clip={clip} ({conv_math}): denoised image max-rel-err {err:.3e}")
        assert err <= 1e-3, (conv_math, err)


def _micro_batches(n, B, seed):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        x0 = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
        Rr = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
        Rr[: B - max(1, B // 8)] = 0                           # ~10 % poisoned rows, the rest clean (R = 0)
        eps = torch.randn(B, 3, 32, 32, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        yield x0, Rr, eps, t


def _update_l2(net, ref, theta0):
    sd = net.state_dict()
    num = sum(float(((sd[k].cpu().double() - v.double()) ** 2).sum()) for k, v in ref.state_dict().items())
    den = sum(float(((v.double() - theta0[k].double()) ** 2).sum()) for k, v in ref.state_dict().items())
    return (num / den) ** 0.5


@pytest.mark.timeout(900)
def test_twenty_optimiser_steps_match_oracle(ref0):
    """20 optimiser steps, batch 8, the reference's hyper-parameters (lr 2e-4, warm-up 500 of 469*50 steps, clip 1.0)."""
    steps, B, lr, warm, total = 20, 8, 2e-4, 500, 469 * 50
    ref = copy.deepcopy(ref0)
    theta0 = {k: v.clone() for k, v in ref.state_dict().items()}
    nets = {cm: _net_like(ref0, cm) for cm in ARITH}
    trs = {cm: Trainer(nets[cm], LossFn(S.DDPMScheduler(), "SDE-VP", psi=1), lr=lr, total_steps=total, warmup_steps=warm) for cm in ARITH}
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: R.cosine_with_warmup_lambda(s, warm, total))
    lf_ref = LossFnRef(R.DDPMSchedulerRef(), SDE_VP, psi=1)
    worst_loss = {cm: 0.0 for cm in ARITH}
    for x0, Rr, eps, t in _micro_batches(steps, B, seed=21):
        l_ref = lf_ref.p_loss(ref, x0, Rr, t, noise=eps)
        l_ref.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        for cm in ARITH:
            assert abs(trs[cm].lr - opt.param_groups[0]["lr"]) < 1e-15
            l = trs[cm].train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=eps.cuda())
            worst_loss[cm] = max(worst_loss[cm], abs(float(l) - float(l_ref)) / abs(float(l_ref)))
        opt.step()
        sch.step()
        opt.zero_grad()
    for cm in ARITH:
        upd = _update_l2(nets[cm], ref, theta0)
        print(f"[parity] 20 optimiser steps ({cm}): worst per-step loss rel-err {worst_loss[cm]:.3e}, update L2 error {upd:.3e}")
        assert trs[cm].opt.step_count == steps and trs[cm].sched_step == steps
        # measured on MI355X: per-step loss 2.4e-7 (bf16x3) / 1.2e-7 (f32), update L2 error 3.5e-5 / 1.2e-5
        assert worst_loss[cm] <= 1e-5, (cm, worst_loss[cm])
        assert upd <= 1e-3, (cm, upd)


@pytest.mark.timeout(900)
def test_config1_batch4_grad_accum_32_one_sync_step(ref0):
    """BASELINE config #1: --batch 4 on a 32x32 dataset -> G = 128 // 4 = 32 micro-steps per optimiser step."""
    G, B, lr, warm, total = 32, 4, 2e-4, 500, 128 * 1
    ref = copy.deepcopy(ref0)
    theta0 = {k: v.clone() for k, v in ref.state_dict().items()}
    nets = {cm: _net_like(ref0, cm) for cm in ARITH}
    trs = {cm: Trainer(nets[cm], LossFn(S.DDPMScheduler(), "SDE-VP", psi=1), lr=lr, total_steps=total, warmup_steps=warm, grad_accum=G)
           for cm in ARITH}
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: R.cosine_with_warmup_lambda(s, warm, total))
    sch.step()                      # LR of sync step 1 (step 0 has lr = 0 under the 500-step warm-up: nothing would move)
    for cm in ARITH:
        trs[cm].sched_step = 1
    lf_ref = LossFnRef(R.DDPMSchedulerRef(), SDE_VP, psi=1)
    for k, (x0, Rr, eps, t) in enumerate(_micro_batches(G, B, seed=33)):
        l_ref = lf_ref.p_loss(ref, x0, Rr, t, noise=eps)
        (l_ref / G).backward()                                  # accelerator.backward divides by G
        for cm in ARITH:
            l = trs[cm].train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=eps.cuda())
            assert abs(float(l) - float(l_ref)) <= 2e-5 * abs(float(l_ref)), (cm, k)
            assert trs[cm].opt.step_count == (1 if k == G - 1 else 0)           # optimiser + LR scheduler move on the sync micro-step only
    gn_ref = float(torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0))
    opt.step()
    for cm in ARITH:
        gn = trs[cm].opt.grad_norm()
        upd = _update_l2(nets[cm], ref, theta0)
        print(f"[parity] G=32 sync step ({cm}): grad-norm rel-err {abs(gn - gn_ref) / gn_ref:.3e}, update L2 error {upd:.3e}")
        assert abs(gn - gn_ref) <= 1e-4 * gn_ref
        # The first Adam step is lr * sign(g) per element: an element whose gradient sits at rounding level flips and contributes a
        # full-size difference whatever the arithmetic.  Measured: 3.3e-4 (bf16x3) / 8.8e-5 (f32)
        assert upd <= 3e-3, (cm, upd)
        assert trs[cm].sched_step == 2


def test_accumulation_boundaries_restart_every_epoch(ref0):
    """accelerate's accumulate(): sync on every G-th micro-step AND on the last batch of the dataloader, where its step counter
    restarts.  3 batches per epoch at G = 2 over two epochs -> sync pattern [0 1 1 | 0 1 1]; the parameters after
    the four optimiser steps are held to the oracle driven by that pattern."""
    G, nb, lr = 2, 3, 1e-3
    ref = copy.deepcopy(ref0)
    theta0 = {k: v.clone() for k, v in ref.state_dict().items()}
    net = _net_like(ref0, "f32")
    tr = Trainer(net, LossFn(S.DDPMScheduler(), "SDE-VP", psi=1), lr=lr, total_steps=100, warmup_steps=0, grad_accum=G)
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    lf_ref = LossFnRef(R.DDPMSchedulerRef(), SDE_VP, psi=1)
    batches = list(_micro_batches(2 * nb, 2, seed=44))
    pattern = []
    for ep in range(2):
        acc_step = 0
        for i in range(nb):
            x0, Rr, eps, t = batches[ep * nb + i]
            end = i == nb - 1
            if end:
                acc_step, sync = 0, True
            else:
                acc_step += 1
                sync = acc_step % G == 0
            (lf_ref.p_loss(ref, x0, Rr, t, noise=eps) / G).backward()
            before = tr.opt.step_count
            tr.train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=eps.cuda(), last_batch=end)
            pattern.append(tr.opt.step_count - before)
            assert pattern[-1] == int(sync), (ep, i, pattern)
            if sync:
                torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
                opt.step()
                opt.zero_grad()
    assert pattern == [0, 1, 1, 0, 1, 1]
    upd = _update_l2(net, ref, theta0)
    print(f"[parity] accumulation across epochs: update L2 error {upd:.3e}")
    assert upd <= 5e-3            # measured 6.5e-4 (four early-Adam steps of a 2-image batch: sign flips of rounding-level gradients)


def test_graphed_micro_step_is_the_eager_micro_step(ref0):
    """Trainer(graph_micro_step=True) replays q-sample -> forward -> loss -> backward of a micro-batch as ONE HIP graph (config #1 is launch-bound
    eagerly: ~470 launches per micro-step).  Same kernels, same order, same fixed-order reductions: the accumulated gradient, the per-micro-step
    losses and the parameters after the optimiser steps must be BIT-identical to the eager launches, across weight updates (the packed
    split-precision operands are refreshed outside the graph) and for the Philox noise drawn when no noise tensor is passed."""
    G, B = 4, 4
    out = {}
    for graphed in (False, True):
        net = _net_like(ref0, "bf16x3")
        lf = LossFn(S.DDPMScheduler(), "SDE-VP", psi=1)
        lf.noise_seed = 77
        tr = Trainer(net, lf, lr=1e-3, total_steps=100, warmup_steps=0, grad_accum=G, graph_micro_step=graphed)
        losses, snaps = [], []
        for k, (x0, Rr, eps, t) in enumerate(_micro_batches(3 * G, B, seed=55)):
            noise = eps.cuda() if k % 2 == 0 else None            # alternate: caller-supplied noise / the loss function's own device RNG
            losses.append(float(tr.train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=noise)))
            if k == G - 2:
                snaps.append(net.flat_grad.clone())               # mid-accumulation gradient
        torch.cuda.synchronize()
        assert tr.opt.step_count == 3 and (len(tr._graphs) == 1) == graphed
        out[graphed] = (losses, snaps, net.flat_param.clone())
    assert out[False][0] == out[True][0]
    assert torch.equal(out[False][1][0], out[True][1][0]) and float(out[True][1][0].abs().max()) > 0
    assert torch.equal(out[False][2], out[True][2])
