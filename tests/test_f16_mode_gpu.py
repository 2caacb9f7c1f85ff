"""Opt-in mixed-precision mode (round 4; the reference trains SDE-VP / SDE-LDM under fp16 autocast + GradScaler, VillanDiffusion.py:260-264, 354):
`conv_math = "f16"` runs the full-size 3x3 / 1x1 forward and input-gradient contractions as ONE f16 product per term on the persistent 16x16x32
kernels (vd_gemm_desc.math = 2), everything else as in the default split-precision arithmetic; the Trainer scales the loss gradient by a power of
two and the Adam kernel skips a step whose gradient is not finite.  Never the headline arithmetic: held here to f16 tolerances against torch f32
and against the default mode."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import A_COL, B_CONV3, B_CONV3_T, B_CONV3_UP, B_PLAIN  # noqa: E402

DEV = "cuda"
F16_TOL = 2e-3          # one f16 rounding per operand: 2^-11 each, random signs over the contraction


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,Cin,Cout,H,mode", [(64, 128, 128, 32, B_CONV3), (128, 256, 256, 16, B_CONV3), (64, 128, 128, 16, B_CONV3_UP), (8, 64, 128, 128, B_CONV3)])
def test_f16_convolution_forward_and_input_gradient(B, Cin, Cout, H, mode):
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g(2))
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == B_CONV3_UP else x
    y0 = F.conv2d(xin, w, b, padding=1)
    res = torch.randn(y0.shape, generator=g(3))
    OH = y0.shape[-1]
    wd = w.to(DEV).view(Cout, -1)
    pk = (ops.conv3_pack_weights(wd, Cout, Cin), ops.conv3_pack_weights_f16(wd, Cout, Cin))
    out = torch.empty(B, Cout, OH, OH, device=DEV)
    ops.conv3x3(x.detach().to(DEV), wd, b.to(DEV), out, mode=mode, residual=res.to(DEV), a_packed=pk)
    assert ops.LAST_GEMM_TILE == 18 and ops.LAST_GEMM_MATH == 2
    e = rel(out, (y0 + res).detach())
    out3 = torch.empty_like(out)
    ops.conv3x3(x.detach().to(DEV), wd, b.to(DEV), out3, mode=mode, residual=res.to(DEV), a_packed=pk[0])
    e3 = rel(out3, (y0 + res).detach())
    print(f"[parity] f16 conv mode={mode} {Cin}->{Cout}@{H}: {e:.2e} (split precision: {e3:.2e})")
    assert e <= F16_TOL and e3 < e                               # the f16 product is what it says: an order of magnitude looser than bf16x3, not broken
    if mode == B_CONV3:
        dy = torch.randn(y0.shape, generator=g(4))
        y0.backward(dy)
        pkt = (ops.conv3_pack_weights(wd, Cin, Cout, transposed=True), ops.conv3_pack_weights_f16(wd, Cin, Cout, transposed=True))
        dx = torch.empty(B, Cin, H, H, device=DEV)
        ops.conv3x3(dy.to(DEV), torch.empty(Cin, Cout * 9, device=DEV), None, dx, mode=B_CONV3_T, a_packed=pkt)
        assert ops.LAST_GEMM_MATH == 2
        assert rel(dx, x.grad) <= F16_TOL
        if H > 32:
            return
        # GroupNorm + SiLU folded into the loader (the sampler's forward) in f16 as well
        gamma, beta = torch.rand(Cin, generator=g(5)) + 0.5, torch.randn(Cin, generator=g(6)) * 0.1
        xd = x.detach().to(DEV)
        a = torch.empty_like(xd)
        mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
        ops.groupnorm_fwd(xd, gamma.to(DEV), beta.to(DEV), a, mean, rstd, 32, 1e-6, True)
        ss = torch.empty(B, Cin, 2, device=DEV)
        ops.groupnorm_stats(xd, gamma.to(DEV), beta.to(DEV), ss, mean, rstd, 32, 1e-6)
        o3, o4 = torch.empty_like(out), torch.empty_like(out)
        ops.conv3x3(xd, wd, b.to(DEV), o3, gn_ss=ss, a_packed=pk)
        assert ops.LAST_GEMM_MATH == 2
        ops.conv3x3(a, wd, b.to(DEV), o4, a_packed=pk)
        assert torch.equal(o3, o4)                               # same f16 operands either way
    # a problem the persistent kernels do not take keeps the split-precision operand
    small = torch.empty(2, Cout, OH, OH, device=DEV)
    ops.conv3x3(x.detach()[:2].to(DEV), wd, b.to(DEV), small, mode=mode, a_packed=pk)
    assert ops.LAST_GEMM_MATH == 0 and rel(small, y0[:2].detach()) <= 1e-4


@pytest.mark.parametrize("B,Cin,Cout,H", [(128, 256, 768, 16), (128, 384, 128, 32)])
def test_f16_1x1_convolution_and_input_gradient(B, Cin, Cout, H):
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = torch.randn(Cout, Cin, 1, 1, generator=g(1)) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g(2))
    wd = w.to(DEV).view(Cout, Cin)
    pk = (ops.conv3_pack_weights(wd, Cout, Cin, taps=1), ops.conv3_pack_weights_f16(wd, Cout, Cin, taps=1))
    out = torch.empty(B, Cout, H, H, device=DEV)
    ops.conv1x1(x.to(DEV), wd, b.to(DEV), out, a_packed=pk)
    assert ops.LAST_GEMM_TILE == 19 and ops.LAST_GEMM_MATH == 2
    assert rel(out, F.conv2d(x, w, b)) <= F16_TOL
    dy = torch.randn(B, Cout, H, H, generator=g(4))
    pkt = (ops.conv3_pack_weights(wd, Cin, Cout, transposed=True, taps=1), ops.conv3_pack_weights_f16(wd, Cin, Cout, transposed=True, taps=1))
    dx = torch.empty(B, Cin, H, H, device=DEV)
    HW = H * H
    ops.gemm(wd, dy.to(DEV), dx, M=Cin, N=B * HW, K=Cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=Cin, ldb=HW, b_bstride=Cout * HW, ldd=HW,
             d_bstride=Cin * HW, a_packed=pkt)
    assert ops.LAST_GEMM_MATH == 2
    assert rel(dx, F.conv_transpose2d(dy, w)) <= F16_TOL


def test_f16_training_mode_tracks_the_default_arithmetic():
    """UNet forward / backward at B = 64 in "f16" mode against the default mode (same weights, same inputs): output and gradient within f16
    tolerances; the loss scale is exact (a power of two multiplied in and divided out in f32); a non-finite gradient skips the Adam step."""
    from villandiffusion_amd.loss import LossFn
    from villandiffusion_amd.schedulers import DDPMScheduler
    from villandiffusion_amd.trainer import Trainer
    from villandiffusion_amd.unet import UNet2DModel
    net = UNet2DModel()
    net.reset_parameters(seed=3)
    B = 64
    x = torch.randn(B, 3, 32, 32, generator=g(1)).cuda()
    t = torch.randint(0, 1000, (B,), generator=g(2)).cuda()
    dy = torch.randn(B, 3, 32, 32, generator=g(3)).cuda() * 1e-4          # gradient magnitudes of a mean-reduced loss
    res = {}
    for mode, scale in (("bf16x3", 1.0), ("f16", 4096.0), ("f16", 1.0)):
        net.conv_math = mode
        net.zero_grad()
        y = net(x, t, return_dict=False)[0]
        y.backward(dy * scale)
        torch.cuda.synchronize()
        res[(mode, scale)] = (y.detach().clone(), net.flat_grad.detach().clone() / scale)
    y3, g3 = res[("bf16x3", 1.0)]
    y16, g16 = res[("f16", 4096.0)]
    ey, eg = rel(y16, y3), float((g16 - g3).norm() / g3.norm())
    print(f"[parity] f16 mode vs default: output {ey:.2e}, gradient (L2) {eg:.2e}; unscaled f16 gradient (L2) "
          f"{float((res[('f16', 1.0)][1] - g3).norm() / g3.norm()):.2e}")
    assert ey <= 1e-2 and eg <= 3e-2
    # without the loss scale the 1e-4-sized input gradients lose bits in f16 (denormals): the scaled run must not be worse
    assert eg <= float((res[("f16", 1.0)][1] - g3).norm() / g3.norm()) * 1.05
    # trainer: scaled loss gradient, unscaled update; an overflow skips the step
    net.conv_math = "f16"
    sched = DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, clip_sample=False)
    lf = LossFn(sched, "SDE-VP", psi=1, solver_type="sde")
    tr = Trainer(net, lf, lr=1e-4, total_steps=100, warmup_steps=0)
    batch = {"target": torch.rand(B, 3, 32, 32, generator=g(5)).cuda() * 2 - 1, "pixel_values": torch.zeros(B, 3, 32, 32, device=DEV)}
    l0 = float(tr.train_step(batch, t))
    assert lf.grad_scale == tr.loss_scale == 4096.0 and math.isfinite(l0)
    p_before = net.flat_param.detach().clone()
    l1 = float(tr.train_step(batch, t))
    assert math.isfinite(l1) and not torch.equal(net.flat_param, p_before)
    gn = tr.opt.grad_norm(1.0 / tr.loss_scale)
    assert 0 < gn < 1e3
    # overflow: poison the gradient after the backward pass by hand through the optimiser interface
    p_before = net.flat_param.detach().clone()
    net.flat_grad[123] = float("inf")
    tr.opt.step(lr=1e-4, grad_inv_scale=1.0 / tr.loss_scale, need_norm=True)
    torch.cuda.synchronize()
    assert torch.equal(net.flat_param, p_before)
    steps_before, sched_before = tr.opt.step_count, tr.sched_step
    tr.check_skipped(force=True)                               # the lazy check (every `scale_check_every` steps in a run)
    assert tr.loss_scale == 2048.0 and tr.overflow_steps_seen == 1
    assert tr._recheck                                         # a check that found skipped steps is repeated after EVERY step until one passes clean
    assert tr.opt.step_count == steps_before - 1               # the refused step does not count for the bias correction ...
    assert tr.sched_step == max(0, sched_before - 1)           # ... nor for the LR schedule (GradScaler / accelerate skip both)
    net.zero_grad()
    # the NEXT step runs under the new scale (advisor r4: the scale used to be re-read once per epoch only) and moves the parameters
    p_before = net.flat_param.detach().clone()
    l2 = float(tr.train_step(batch, t))
    assert lf.grad_scale == 2048.0 / tr.grad_accum and tr._step_scale == 2048.0
    assert math.isfinite(l2) and not torch.equal(net.flat_param, p_before)
    assert not tr._recheck                                     # (that step was clean: back to the `scale_check_every` cadence)
    # the scale state survives a checkpoint; saving one reads the skip counter but is NOT an optimiser step (advisor r5: it used to advance the
    # growth interval, so a checkpoint every few steps shortened the 2000-step interval)
    growth_before, scale_before = tr._since_growth, tr.loss_scale
    sd = tr.state_dict()
    sd = tr.state_dict()
    assert tr._since_growth == growth_before and tr.loss_scale == scale_before and sd["since_growth"] == growth_before
    tr2 = Trainer(net, lf, lr=1e-4, total_steps=100, warmup_steps=0)
    tr2.load_state_dict(sd)
    assert tr2.loss_scale == 2048.0 and tr2.overflow_steps_seen == 1
    net.zero_grad()
    net.conv_math = "bf16x3"
    # default arithmetic: there is no scale to lower -- a non-finite gradient norm is refused loudly at the check
    tr3 = Trainer(net, lf, lr=1e-4, total_steps=100, warmup_steps=0)
    p_before = net.flat_param.detach().clone()
    net.flat_grad[7] = float("nan")
    tr3.opt.step(lr=1e-4)
    torch.cuda.synchronize()
    assert torch.equal(net.flat_param, p_before)
    import pytest as _pytest
    with _pytest.raises(FloatingPointError):
        tr3.check_skipped(force=True)
    net.zero_grad()
