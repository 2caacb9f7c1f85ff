"""Known-answer / identity tests for the (parity-unpinned) sampler oracle, SURVEY.md §8c."""
import numpy as np
import torch

from oracle import schedulers_ref as R

from oracle.schedulers_ref import (DDIMSchedulerRef, DDPMSchedulerRef, DPMSolverMultistepSchedulerRef,
                                   ScoreSdeVeSchedulerRef, UniPCMultistepSchedulerRef, cosine_with_warmup_lambda)

DPM20 = [999, 949, 899, 849, 799, 749, 699, 649, 599, 549, 500, 450, 400, 350, 300, 250, 200, 150, 100, 50]


def test_timestep_tables_bit_exact():
    s = DDPMSchedulerRef(); s.set_timesteps(1000)
    assert s.timesteps.tolist() == list(range(999, -1, -1)) and s.timesteps.dtype == torch.int64
    s = DDIMSchedulerRef(); s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    for cls in (DPMSolverMultistepSchedulerRef, UniPCMultistepSchedulerRef):
        s = cls(); s.set_timesteps(20)
        assert s.timesteps.tolist() == DPM20


def _eps_model(x, t):
    return 0.3 * x + 0.05 * torch.cos(x * 3.0 + float(t) * 1e-3)


def _run(s, n, x, **kw):
    s.set_timesteps(n)
    for t in s.timesteps:
        x = s.step(_eps_model(x, t), t, x, **kw).prev_sample
    return x


def test_dpmpp_order1_equals_ddim_eta0():
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    dpm = DPMSolverMultistepSchedulerRef(solver_order=1)
    dpm.set_timesteps(20)
    ddim = DDIMSchedulerRef(clip_sample=False)
    # walk DDIM over the DPM timestep grid by hand (same t -> prev_t pairs)
    xa = x.clone(); xb = x.clone()
    ts = dpm.timesteps.tolist()
    for i, t in enumerate(ts):
        prev = 0 if i == len(ts) - 1 else ts[i + 1]
        xa = dpm.step(_eps_model(xa, t), t, xa).prev_sample
        a_t, a_p = ddim.alphas_cumprod[t], ddim.alphas_cumprod[prev]
        e = _eps_model(xb, t)
        x0 = (xb - (1 - a_t) ** 0.5 * e) / a_t ** 0.5
        xb = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e
    assert torch.allclose(xa, xb, rtol=1e-4, atol=1e-5)


def test_unipc_first_step_equals_ddim():
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(1))
    u = UniPCMultistepSchedulerRef(); u.set_timesteps(20)
    t, prev = 999, 949
    e = _eps_model(x, t)
    xa = u.step(e, t, x).prev_sample
    ac = u.alphas_cumprod
    x0 = (x - (1 - ac[t]) ** 0.5 * e) / ac[t] ** 0.5
    xb = ac[prev] ** 0.5 * x0 + (1 - ac[prev]) ** 0.5 * e
    assert torch.allclose(xa, xb, rtol=1e-4, atol=1e-5)


def test_ddim_eta1_equals_ddpm_posterior():
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(2))
    z = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(3))
    d, i = DDPMSchedulerRef(clip_sample=False), DDIMSchedulerRef(clip_sample=False)
    d.set_timesteps(1000); i.set_timesteps(1000)
    for t in (999, 500, 3):
        e = _eps_model(x, t)
        a = d.step(e, t, x, noise=z).prev_sample
        b = i.step(e, t, x, eta=1.0, noise=z).prev_sample
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4), t


def test_ddpm_t0_is_deterministic_and_clips():
    d = DDPMSchedulerRef(clip_sample=True); d.set_timesteps(1000)
    x = torch.full((1, 3, 4, 4), 5.0)
    out = d.step(torch.zeros_like(x), 0, x)
    assert float(out.pred_original_sample.max()) == 1.0
    assert torch.equal(out.prev_sample, d.step(torch.zeros_like(x), 0, x).prev_sample)


def test_add_noise_matches_closed_form():
    d = DDPMSchedulerRef()
    x0, e = torch.randn(3, 3, 4, 4), torch.randn(3, 3, 4, 4)
    t = torch.tensor([0, 400, 999])
    ref = torch.stack([d.alphas_cumprod[k] ** 0.5 * x0[j] + (1 - d.alphas_cumprod[k]) ** 0.5 * e[j] for j, k in enumerate(t)])
    assert torch.equal(d.add_noise(x0, e, t), ref)


def _run_gauss(s, n, x, std=0.5):
    """Exact eps for data ~ N(0, std^2): every consistent ODE solver converges to x_T * std (as T->0)."""
    s.set_timesteps(n)
    for t in s.timesteps:
        ac = s.alphas_cumprod[int(t)]
        eps = (1 - ac) ** 0.5 * x / (ac * std ** 2 + (1 - ac))
        x = s.step(eps, t, x).prev_sample
    return x


def test_dpm_unipc_convergence_order():
    """On Gaussian data the probability-flow ODE is linear, so the exact answer is known:
    order-p solvers must shrink their error ~2^p per step-doubling (measured 100 -> 200 steps)."""
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(4))
    ac = DDPMSchedulerRef().alphas_cumprod
    exact = x * float((ac[0] * 0.25 + 1 - ac[0]) ** 0.5 / (ac[999] * 0.25 + 1 - ac[999]) ** 0.5)

    def err(make, n):
        out = _run_gauss(make(), n, x.clone())
        return float((out - exact).abs().max() / exact.abs().max())

    mk = {
        "pp1": (lambda: DPMSolverMultistepSchedulerRef(solver_order=1), 1.7, 2.4),
        "o1": (lambda: DPMSolverMultistepSchedulerRef(solver_order=1, algorithm_type="dpmsolver"), 1.7, 2.4),
        "pp2": (lambda: DPMSolverMultistepSchedulerRef(solver_order=2), 3.2, 5.5),
        "o2": (lambda: DPMSolverMultistepSchedulerRef(solver_order=2, algorithm_type="dpmsolver"), 3.2, 5.5),
        "pp3": (lambda: DPMSolverMultistepSchedulerRef(solver_order=3), 5.0, 1e9),
    }
    for name, (make, lo, hi) in mk.items():
        e100, e200 = err(make, 100), err(make, 200)
        assert lo < e100 / e200 < hi, (name, e100, e200)
    assert err(lambda: UniPCMultistepSchedulerRef(), 50) < 3e-3
    assert err(lambda: DPMSolverMultistepSchedulerRef(solver_order=3, algorithm_type="dpmsolver"), 50) < 3e-3
    # 20-step headline configs stay finite and in the right ballpark
    for make in (mk["pp2"][0], lambda: UniPCMultistepSchedulerRef()):
        assert err(make, 20) < 0.3


def test_score_sde_ve_tables():
    s = ScoreSdeVeSchedulerRef(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    assert s.sigmas.shape == (2000,) and float(s.sigmas[0]) == 380.0
    assert abs(float(s.sigmas[-1]) - 0.0100010550) < 1e-8
    s.set_timesteps(10); s.set_sigmas(10)
    x = torch.randn(2, 3, 4, 4)
    out = s.step_pred(-x / 100, s.timesteps[0], x, noise=torch.zeros_like(x))
    assert torch.isfinite(out.prev_sample).all()
    out = s.step_correct(-x, x, noise=torch.ones_like(x))
    assert torch.isfinite(out.prev_sample).all()


def test_cosine_warmup():
    assert cosine_with_warmup_lambda(0, 500, 23450) == 0.0
    assert cosine_with_warmup_lambda(250, 500, 23450) == 0.5
    assert cosine_with_warmup_lambda(500, 500, 23450) == 1.0
    assert abs(cosine_with_warmup_lambda(23450, 500, 23450)) < 1e-12
    mid = 500 + (23450 - 500) // 2
    assert abs(cosine_with_warmup_lambda(mid, 500, 23450) - 0.5) < 1e-3


def test_pndm_timestep_table_and_next_row_sampler_convergence():
    """§8f.3 samplers.  PNDM table for (T=1000, n=50): 12 Runge-Kutta calls at 980..920 then PLMS down to 0 (59 UNet calls);
    all of PNDM / DEIS / Heun / LMSD must converge to the exact probability-flow solution of a Gaussian data model."""
    s = R.PNDMSchedulerRef()
    s.set_timesteps(50)
    assert s.timesteps[:13].tolist() == [980, 970, 970, 960, 960, 950, 950, 940, 940, 930, 930, 920, 920]
    assert len(s.timesteps) == 59 and s.timesteps[-1] == 0 and s.timesteps.dtype == torch.int64
    h = R.HeunDiscreteSchedulerRef()
    h.set_timesteps(10)
    assert len(h.timesteps) == 19 and len(h.sigmas) == 20 and float(h.sigmas[-1]) == 0.0
    assert abs(float(h.init_noise_sigma) - float(((1 - h.alphas_cumprod[-1]) / h.alphas_cumprod[-1]) ** 0.5)) < 1e-4

    ac = R._VPBase().alphas_cumprod.double()
    sg_tab = ((1 - ac) / ac) ** 0.5
    v0 = 0.25                                             # data ~ N(0, v0 I): eps*(x_vp, abar) = sqrt(1-abar) x / (abar v0 + 1 - abar)

    def run(sc, n, x):
        sc.set_timesteps(n)
        sig_space = hasattr(sc, "sigmas")
        if sig_space:
            x = x * sc.init_noise_sigma
        for t in sc.timesteps:
            tf = float(t)
            lo = int(np.floor(tf)); hi = min(lo + 1, 999); w = tf - lo
            sig = sg_tab[lo] * (1 - w) + sg_tab[hi] * w
            a = 1 / (sig ** 2 + 1)
            xin = sc.scale_model_input(x, t) if sig_space else x
            x = sc.step(((1 - a) ** 0.5 * xin / (a * v0 + 1 - a)).float(), t, x).prev_sample
        return x

    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    sm = float(sg_tab[999])
    exact_vp = x * float(((v0 * ac[0] + 1 - ac[0]) / (v0 * ac[999] + 1 - ac[999])) ** 0.5)
    exact_sigma = x * sm * (v0 / (v0 + sm ** 2)) ** 0.5
    for mk, exact in [(R.PNDMSchedulerRef, exact_vp), (R.DEISMultistepSchedulerRef, exact_vp),
                      (R.HeunDiscreteSchedulerRef, exact_sigma), (R.LMSDiscreteSchedulerRef, exact_sigma)]:
        errs = [float((run(mk(), n, x.clone()) - exact).abs().max() / exact.abs().max()) for n in (20, 80)]
        assert errs[1] < errs[0] and errs[1] < 6e-3, (mk.__name__, errs)
