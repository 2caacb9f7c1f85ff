"""Host logic of the product samplers (timestep tables, per-step scalar coefficients, multistep bookkeeping) against
the oracle.  No GPU here: the two HIP entry points the samplers call are replaced by torch restatements of what the
kernels compute (csrc/vd_elem.hip), so only the host side is under test; the kernels themselves are checked in
tests/test_hip_kernels.py (-m gpu)."""
import pytest
import torch

from oracle import schedulers_ref as R
from villandiffusion_amd import ops, schedulers as S


def _fake_sched_step(x, eps, out, *, c_eps, c_div, clip, c_x0, c_x, c_e, c_z, z=None, x0_out=None, seed=0, offset=0):
    f = lambda v: torch.tensor(v, dtype=torch.float32)
    x0 = (x - f(c_eps) * eps) / f(c_div)
    if clip > 0:
        x0 = x0.clamp(-clip, clip)
    o = f(c_x0) * x0 + f(c_x) * x
    if c_e != 0:
        o = o + f(c_e) * eps
    if c_z != 0:
        o = o + f(c_z) * z
    out.copy_(o)
    if x0_out is not None:
        x0_out.copy_(x0)
    return out


def _fake_lincomb(out, srcs, coefs):
    acc = torch.tensor(coefs[0], dtype=torch.float32) * srcs[0]
    for s, c in zip(srcs[1:], coefs[1:]):
        acc = acc + torch.tensor(c, dtype=torch.float32) * s
    out.copy_(acc)
    return out


def _fake_batch_l2norm(x, out):
    out[:x.shape[0]].copy_(torch.norm(x.reshape(x.shape[0], -1), dim=-1))
    return out


@pytest.fixture(autouse=True)
def _patch(monkeypatch):
    monkeypatch.setattr(ops, "sched_step", _fake_sched_step)
    monkeypatch.setattr(ops, "lincomb", _fake_lincomb)
    monkeypatch.setattr(ops, "batch_l2norm", _fake_batch_l2norm)


def _eps(x, t, ac):
    a = ac[int(t)]
    return (1 - a) ** 0.5 * x / (a * 0.25 + (1 - a)) + 0.01 * torch.sin(3 * x)


def _run(s, n, x, **kw):
    s.set_timesteps(n)
    for t in s.timesteps:
        x = s.step(_eps(x, t, s.alphas_cumprod), t, x, **kw).prev_sample
    return x


PAIRS = [
    (lambda: S.DDPMScheduler(clip_sample=False), lambda: R.DDPMSchedulerRef(clip_sample=False), 1000, 50),
    (lambda: S.DDPMScheduler(clip_sample=True), lambda: R.DDPMSchedulerRef(clip_sample=True), 1000, 50),
    (lambda: S.DDIMScheduler(clip_sample=False), lambda: R.DDIMSchedulerRef(clip_sample=False), 50, 50),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=1), lambda: R.DPMSolverMultistepSchedulerRef(solver_order=1), 20, 20),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=2), lambda: R.DPMSolverMultistepSchedulerRef(solver_order=2), 20, 20),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=3), lambda: R.DPMSolverMultistepSchedulerRef(solver_order=3), 20, 20),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=2, algorithm_type="dpmsolver"),
     lambda: R.DPMSolverMultistepSchedulerRef(solver_order=2, algorithm_type="dpmsolver"), 20, 20),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=3, algorithm_type="dpmsolver"),
     lambda: R.DPMSolverMultistepSchedulerRef(solver_order=3, algorithm_type="dpmsolver"), 20, 20),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=2, solver_type="heun"),
     lambda: R.DPMSolverMultistepSchedulerRef(solver_order=2, solver_type="heun"), 20, 20),
    (lambda: S.DPMSolverMultistepScheduler(solver_order=3), lambda: R.DPMSolverMultistepSchedulerRef(solver_order=3), 10, 10),
    (lambda: S.UniPCMultistepScheduler(), lambda: R.UniPCMultistepSchedulerRef(), 20, 20),
    (lambda: S.UniPCMultistepScheduler(solver_order=3), lambda: R.UniPCMultistepSchedulerRef(solver_order=3), 20, 20),
    (lambda: S.UniPCMultistepScheduler(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear"),
     lambda: R.UniPCMultistepSchedulerRef(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear"), 20, 20),
    # SURVEY §8f.3 samplers (reference model.py:641-652)
    (lambda: S.PNDMScheduler(), lambda: R.PNDMSchedulerRef(), 50, 100),
    (lambda: S.PNDMScheduler(skip_prk_steps=True), lambda: R.PNDMSchedulerRef(skip_prk_steps=True), 20, 100),
    (lambda: S.DEISMultistepScheduler(), lambda: R.DEISMultistepSchedulerRef(), 20, 20),
    (lambda: S.DEISMultistepScheduler(solver_order=3), lambda: R.DEISMultistepSchedulerRef(solver_order=3), 20, 20),
    (lambda: S.DEISMultistepScheduler(solver_order=3), lambda: R.DEISMultistepSchedulerRef(solver_order=3), 10, 10),
    (lambda: S.HeunDiscreteScheduler(), lambda: R.HeunDiscreteSchedulerRef(), 20, 100),
    (lambda: S.LMSDiscreteScheduler(), lambda: R.LMSDiscreteSchedulerRef(), 20, 100),
    (lambda: S.LMSDiscreteScheduler(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear"),
     lambda: R.LMSDiscreteSchedulerRef(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear"), 10, 100),
]


@pytest.mark.parametrize("mk,mkref,n,steps", PAIRS)
def test_sampler_host_logic_matches_oracle(mk, mkref, n, steps):
    a, b = mk(), mkref()
    a.set_timesteps(n); b.set_timesteps(n)
    assert torch.equal(a.timesteps, b.timesteps) and a.timesteps.dtype == b.timesteps.dtype      # bit-exact tables
    sigma_space = hasattr(b, "sigmas")
    assert sigma_space or a.timesteps.dtype == torch.int64
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod)
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    if sigma_space:
        assert torch.equal(a.sigmas, b.sigmas) and float(a.init_noise_sigma) == float(b.init_noise_sigma)
        x = x * b.init_noise_sigma
    xa, xb = x.clone(), x.clone()
    ga, gb = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    for t in a.timesteps[:steps]:
        ia = a.scale_model_input(xa, t) if sigma_space else xa
        ib = b.scale_model_input(xb, t) if sigma_space else xb
        xa = a.step(_eps(ia, t, a.alphas_cumprod), t, xa, generator=ga).prev_sample
        xb = b.step(_eps(ib, t, b.alphas_cumprod), t, xb, generator=gb).prev_sample
    err = float((xa - xb).abs().max() / xb.abs().max())
    assert err < 2e-5, err


def test_ddim_eta_and_generator_stream():
    a, b = S.DDIMScheduler(clip_sample=True), R.DDIMSchedulerRef(clip_sample=True)
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    ga, gb = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    xa = _run(a, 50, x.clone(), eta=0.5, generator=ga)
    xb = _run(b, 50, x.clone(), eta=0.5, generator=gb)
    assert float((xa - xb).abs().max()) < 1e-5


def test_add_noise_and_surface():
    s = S.DDPMScheduler()
    assert s.config.num_train_timesteps == 1000 and s.betas.dtype == torch.float32 and s.betas.device.type == "cpu"
    s.config.clip_sample = False          # settable (model.py:661-663)
    x0, e = torch.randn(3, 3, 4, 4), torch.randn(3, 3, 4, 4)
    t = torch.tensor([0, 400, 999])
    assert torch.equal(s.add_noise(x0, e, t), R.DDPMSchedulerRef().add_noise(x0, e, t))
    lam = S.get_cosine_schedule_with_warmup_lambda(500, 23450)
    assert [lam(0), lam(250), lam(500)] == [0.0, 0.5, 1.0] and abs(lam(23450)) < 1e-12
    for k in (0, 100, 600, 12000, 23449):
        assert lam(k) == R.cosine_with_warmup_lambda(k, 500, 23450)


def test_score_sde_ve_host_logic_matches_oracle():
    a = S.ScoreSdeVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    b = R.ScoreSdeVeSchedulerRef(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    assert torch.equal(a.sigmas, b.sigmas) and torch.equal(a.discrete_sigmas, b.discrete_sigmas)
    a.set_timesteps(30); a.set_sigmas(30); b.set_timesteps(30); b.set_sigmas(30)
    assert torch.equal(a.timesteps, b.timesteps) and torch.equal(a.sigmas, b.sigmas)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 3, 8, 8, generator=g) * 380
    xa, xb = x.clone(), x.clone()
    for i, t in enumerate(a.timesteps):
        z1, z2 = torch.randn(x.shape, generator=g), torch.randn(x.shape, generator=g)
        sig = a.sigmas[i]
        score = lambda v: -v / (sig ** 2 + 0.25)
        xa = a.step_correct(score(xa), xa, noise=z1).prev_sample
        xb = b.step_correct(score(xb), xb, noise=z1).prev_sample
        oa, ob = a.step_pred(score(xa), t, xa, noise=z2), b.step_pred(score(xb), t, xb, noise=z2)
        xa, xb = oa.prev_sample, ob.prev_sample
    err = float((oa.prev_sample_mean - ob.prev_sample_mean).abs().max() / ob.prev_sample_mean.abs().max())
    assert err < 1e-4, err


def test_karras_ve_host_logic_matches_oracle(monkeypatch):
    """KarrasVeScheduler (model.py:685-693): sigma table, churn noise injection, Euler step and 2nd-order correction."""
    for churn in (80.0, 100.0, 0.0):
        a = S.KarrasVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, s_churn=churn)
        b = R.KarrasVeSchedulerRef(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, s_churn=churn)
        a.set_timesteps(12); b.set_timesteps(12)
        assert torch.equal(a.timesteps, b.timesteps) and torch.equal(a.schedule, b.schedule)
        assert a.init_noise_sigma == b.init_noise_sigma == 380.0
        f = lambda x, s: -x / (0.25 + float(s) ** 2) ** 0.5          # stand-in score network
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(0)) * 380.0
        xa, xb = x.clone(), x.clone()
        ga, gb = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
        for t in a.timesteps.tolist():
            sg, sp = a.schedule[t], (a.schedule[t - 1] if t > 0 else 0)
            ha, sha = a.add_noise_to_input(xa, sg, generator=ga)
            hb, shb = b.add_noise_to_input(xb, sg, generator=gb)
            assert float(sha) == float(shb)
            oa, ob = a.step(f(ha, sha), sha, sp, ha), b.step(f(hb, shb), shb, sp, hb)
            if sp != 0:
                oa = a.step_correct(f(oa.prev_sample, sp), sha, sp, ha, oa.prev_sample, oa.derivative)
                ob = b.step_correct(f(ob.prev_sample, sp), shb, sp, hb, ob.prev_sample, ob.derivative)
            xa, xb = oa.prev_sample, ob.prev_sample
        assert float((xa - xb).abs().max() / xb.abs().max()) < 2e-5
