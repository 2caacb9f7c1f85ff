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


# ---- pinned by the reference's own loss.NoiseScheduler / q_sample_clean / q_sample_backdoor (loss.py:62-196), which import and run
# ---- in the build container: tests/golden/noise_scheduler.npz (generated by tests/golden/make_golden.py)
def _ns():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "noise_scheduler.npz"))


def test_beta_tables_match_reference_noise_scheduler():
    """beta / alpha / alpha-bar of the oracle AND the product schedulers == the reference's tables, bit for bit (linear is the VP
    recipe, model.py:606-608; the reference's 'quadratic' schedule is diffusers' scaled_linear rule at the same end points)."""
    from villandiffusion_amd import schedulers as S
    ns = _ns()
    for tag, kw in (("linear", dict(beta_schedule="linear")), ("quadratic", dict(beta_schedule="scaled_linear"))):
        for mk in (DDPMSchedulerRef, S.DDPMScheduler, S.DPMSolverMultistepScheduler, S.UniPCMultistepScheduler):
            s = mk(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, **kw)
            assert np.array_equal(s.betas.numpy(), ns[f"{tag}/betas"]), (tag, mk)
            assert np.array_equal(s.alphas.numpy(), ns[f"{tag}/alphas"])
            assert np.array_equal(s.alphas_cumprod.numpy(), ns[f"{tag}/alphas_cumprod"])


def test_ddpm_variance_table_matches_reference_posterior_variance():
    """The reference's posterior variance is beta_t (1 - abar_{t-1}) / (1 - abar_t); upstream DDPMScheduler (restated by the oracle and
    the product) takes beta_t as 1 - abar_t / abar_{t-1}, which carries the fp32 rounding of the abar ratio: relative 6e-8 / beta_t,
    i.e. up to ~6e-4 at the small-beta end.  Held to 2e-3 relative over all 1000 steps; t = 0 is exactly 0 in both."""
    from villandiffusion_amd import schedulers as S
    ns = _ns()
    pv = ns["linear/posterior_variance"].astype(np.float64)
    for s in (DDPMSchedulerRef(), S.DDPMScheduler()):
        ac, one = s.alphas_cumprod, torch.tensor(1.0)
        var = []
        for t in range(1000):
            a_t, a_prev = ac[t], (ac[t - 1] if t > 0 else one)
            var.append(float((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)))
        var = np.array(var)
        assert var[0] == 0.0 and pv[0] == 0.0
        assert np.abs(var[1:] - pv[1:]).max() / 1.0 <= 1e-6 and (np.abs(var[1:] - pv[1:]) / pv[1:]).max() <= 2e-3
        # fixed_small noise scale of the scheduler under test == sqrt of that table (clamped at 1e-20), fixed_large == sqrt(beta_t)
    d_small, d_large = DDPMSchedulerRef(), DDPMSchedulerRef(variance_type="fixed_large")
    p_small, p_large = S.DDPMScheduler(), S.DDPMScheduler(variance_type="fixed_large")
    for t in (1, 2, 10, 500, 999):
        a_t, a_prev = d_small.alphas_cumprod[t], d_small.alphas_cumprod[t - 1]
        cb = 1 - a_t / a_prev
        assert abs(float(d_small.noise_scale(a_t, a_prev, cb)) - pv[t] ** 0.5) <= 1e-3 * pv[t] ** 0.5
        assert abs(float(d_large.noise_scale(a_t, a_prev, cb)) - float(ns["linear/betas"][t]) ** 0.5) <= 1e-3 * float(ns["linear/betas"][t]) ** 0.5
        assert p_small._noise_scale(a_t, a_prev, cb) == float(d_small.noise_scale(a_t, a_prev, cb))
        assert p_large._noise_scale(a_t, a_prev, cb) == float(d_large.noise_scale(a_t, a_prev, cb))
    assert abs(float(d_small.noise_scale(torch.tensor(0.5), torch.tensor(1.0), torch.tensor(0.5))) - 1e-10) < 1e-16        # the 1e-20 clamp


def test_add_noise_and_backdoor_qsample_match_reference():
    """`add_noise` (oracle and product) == loss.q_sample_clean; the psi = 1 / 'sde' inputs and targets of the oracle's LossFnRef ==
    loss.q_sample_backdoor, on the seeded batch of tests/golden/loss_batch.npz (t = 0, 10, 500, 999).  The reference gathers
    torch.sqrt tables, the schedulers raise to 0.5 and the loss builds 1 - sqrt(abar) differently: equal to 1 ulp of the terms."""
    import os
    from oracle.loss_ref import LossFnRef, SDE_VP
    from villandiffusion_amd import schedulers as S
    ns = _ns()
    b = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_batch.npz"))
    x0, Rr, eps, t = (torch.from_numpy(b[k]) for k in ("x0", "R", "eps", "t_vp"))
    for tag, kw in (("linear", dict(beta_schedule="linear")), ("quadratic", dict(beta_schedule="scaled_linear"))):
        clean = torch.from_numpy(ns[f"{tag}/q_sample_clean"])
        for s in (DDPMSchedulerRef(**kw), S.DDPMScheduler(**kw), S.DDIMScheduler(**kw)):
            assert torch.allclose(s.add_noise(x0, eps, t), clean, rtol=0, atol=5e-7), tag
        lf = LossFnRef(DDPMSchedulerRef(**kw), SDE_VP, psi=1, solver_type="sde")
        xt, y = lf.inputs_targets(x0, Rr, t, eps)
        assert torch.allclose(xt, torch.from_numpy(ns[f"{tag}/q_sample_backdoor_x"]), rtol=0, atol=1e-6)
        assert torch.allclose(y, torch.from_numpy(ns[f"{tag}/q_sample_backdoor_y"]), rtol=0, atol=1e-6)
        step, coef = lf.tables()
        assert np.abs(coef.numpy() - ns[f"{tag}/R_coef"]).max() <= 1e-6          # BadDiffusion R coefficient (loss.py:103)
