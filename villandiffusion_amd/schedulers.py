"""Reverse-diffusion samplers (S1-S5 of SURVEY.md §8a, plus the §8f.3 ones: PNDM, DEIS, Heun, LMSD, KarrasVe) with the
per-step tensor update on the GPU.

Same surface the reference touches on its diffusers schedulers (model.py:599-665, loss.py:830-834,
VillanDiffusion.py:1151): ``.betas/.alphas/.alphas_cumprod`` (fp32, CPU), ``.config.num_train_timesteps``,
``.config.clip_sample`` (settable), ``.set_timesteps(n)``, ``.timesteps`` (int64, bit-exact upstream rule),
``.add_noise(x0, eps, t)``, ``.step(eps_hat, t, x, generator=..., eta=...) -> .prev_sample``.

Split of work: the scalar coefficients of a step are computed on the host with the SAME fp32 torch op sequence
upstream uses (so they agree with the CPU oracle to the last bit); the update of the [B,C,H,W] state is ONE fused HIP
kernel (``vd_sched_step`` for DDPM/DDIM, ``vd_lincomb`` for the multistep solvers, whose history terms are folded
into per-tensor scalar coefficients on the host).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from . import ops


def make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(f"beta_schedule {beta_schedule}")


def _noise_like(x: torch.Tensor, generator: Optional[torch.Generator]):
    """diffusers randn_tensor: a CPU generator draws on the CPU (reference: VillanDiffusion.py:621-624 seeds a CPU
    generator), then the noise is moved to the device."""
    if generator is not None and generator.device.type == "cpu":
        return torch.randn(x.shape, generator=generator, dtype=x.dtype).to(x.device)
    return torch.randn(x.shape, generator=generator, device=x.device, dtype=x.dtype)


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)


class SchedulerBase:
    _class_name = "SchedulerBase"
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, clip_sample_range=1.0, trained_betas=None, prediction_type="epsilon", **extra):
        if prediction_type != "epsilon":     # the reference only ever builds epsilon-prediction schedulers (model.py:606-652)
            raise NotImplementedError(f"{type(self).__name__}: prediction_type '{prediction_type}' (only 'epsilon' is implemented)")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                              trained_betas=trained_betas, prediction_type="epsilon", **extra)
        self.betas = (torch.tensor(trained_betas, dtype=torch.float32) if trained_betas is not None
                      else make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule))
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        # throughput mode: per-step noise from the in-kernel Philox stream instead of torch.randn
        self.device_rng_seed: Optional[int] = None
        self._rng_offset = 0

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, x0, noise, timesteps):
        ac = self.alphas_cumprod.to(device=x0.device, dtype=x0.dtype)
        t = timesteps.to(x0.device)
        sa, sb = (ac[t] ** 0.5).flatten(), ((1 - ac[t]) ** 0.5).flatten()
        while sa.dim() < x0.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * x0 + sb * noise

    def scheduler_config(self) -> dict:
        d = {k: v for k, v in vars(self.config).items()}
        d["_class_name"] = self._class_name
        return d

    def _noise_args(self, x, generator, noise, needed: bool):
        """(z tensor or None, seed, offset) for vd_sched_step."""
        if not needed:
            return None, 0, 0
        if noise is not None:
            return noise, 0, 0
        if self.device_rng_seed is not None:
            off = self._rng_offset
            self._rng_offset += (x.numel() + 3) // 4
            return None, int(self.device_rng_seed), off
        return _noise_like(x, generator), 0, 0


class DDPMScheduler(SchedulerBase):
    """S1 -- [UPSTREAM] DDPMScheduler, epsilon prediction.  variance_type as diffusers ~0.16 `_get_variance`: fixed_small (the
    posterior variance clamped at 1e-20; what the reference's --sched DDPM-SCHED builds, model.py:615), fixed_small_log, and
    fixed_large (beta_t: what the hub checkpoint google/ddpm-cifar10-32 ships, i.e. what `--sched` unset samples with,
    model.py:654).  fixed_large_log / learned / learned_range raise: the first is sqrt(log(beta)) = NaN upstream, the others need
    a network that predicts the variance."""
    _class_name = "DDPMScheduler"
    VARIANCE_TYPES = ("fixed_small", "fixed_small_log", "fixed_large")

    def __init__(self, *a, variance_type="fixed_small", **k):
        if variance_type not in self.VARIANCE_TYPES:
            raise NotImplementedError(f"DDPMScheduler: variance_type '{variance_type}' (implemented: {self.VARIANCE_TYPES})")
        super().__init__(*a, variance_type=variance_type, **k)

    def _noise_scale(self, a_t, a_prev, cur_beta) -> float:
        vt = self.config.variance_type
        if vt == "fixed_large":
            return float(cur_beta ** 0.5)
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
        if vt == "fixed_small_log":
            return float(torch.exp(0.5 * torch.log(var)))
        return float(var ** 0.5)

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, generator=None, noise=None, return_dict=True, **_):
        t = int(timestep)
        T = self.config.num_train_timesteps
        n = self.num_inference_steps if self.num_inference_steps else T
        prev_t = t - T // n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        c_x0 = (a_prev ** 0.5 * cur_beta) / b_t
        c_xt = cur_alpha ** 0.5 * b_prev / b_t
        c_z = 0.0
        if t > 0:
            c_z = self._noise_scale(a_t, a_prev, cur_beta)
        z, seed, off = self._noise_args(sample, generator, noise, t > 0)
        out, x0 = torch.empty_like(sample), torch.empty_like(sample)
        ops.sched_step(sample.contiguous(), model_output.contiguous(), out, c_eps=float(b_t ** 0.5), c_div=float(a_t ** 0.5),
                       clip=float(self.config.clip_sample_range) if self.config.clip_sample else 0.0, c_x0=float(c_x0),
                       c_x=float(c_xt), c_e=0.0, c_z=c_z, z=z, x0_out=x0, seed=seed, offset=off)
        return SimpleNamespace(prev_sample=out, pred_original_sample=x0)


class DDIMScheduler(SchedulerBase):
    """S2 -- [UPSTREAM] DDIMScheduler."""
    _class_name = "DDIMScheduler"

    def __init__(self, *a, set_alpha_to_one=True, steps_offset=0, **k):
        super().__init__(*a, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **k)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False, generator=None,
             noise=None, return_dict=True, **_):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = (eta or 0.0) * var ** 0.5
        c_e = (1 - a_prev - std ** 2) ** 0.5
        need_noise = (eta or 0.0) > 0
        z, seed, off = self._noise_args(sample, generator, noise, need_noise)
        out, x0 = torch.empty_like(sample), torch.empty_like(sample)
        ops.sched_step(sample.contiguous(), model_output.contiguous(), out, c_eps=float(b_t ** 0.5), c_div=float(a_t ** 0.5),
                       clip=float(self.config.clip_sample_range) if self.config.clip_sample else 0.0,
                       c_x0=float(a_prev ** 0.5), c_x=0.0, c_e=float(c_e), c_z=float(std) if need_noise else 0.0, z=z,
                       x0_out=x0, seed=seed, offset=off)
        return SimpleNamespace(prev_sample=out, pred_original_sample=x0)


class _Multistep(SchedulerBase):
    def __init__(self, *a, solver_order=2, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, solver_order=solver_order, **k)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.lower_order_nums = 0

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, T - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0

    def _step_index(self, timestep) -> int:
        idx = (self.timesteps == int(timestep)).nonzero()
        return len(self.timesteps) - 1 if len(idx) == 0 else int(idx[0].item())

    def _x0_pred(self, eps, t, sample):
        out = torch.empty_like(sample)
        ops.sched_step(sample.contiguous(), eps.contiguous(), out, c_eps=float(self.sigma_t[t]), c_div=float(self.alpha_t[t]),
                       clip=0.0, c_x0=1.0, c_x=0.0, c_e=0.0, c_z=0.0)
        return out


class DPMSolverMultistepScheduler(_Multistep):
    """S3 -- [UPSTREAM] DPMSolverMultistepScheduler (dpmsolver / dpmsolver++, midpoint, orders 1-3)."""
    _class_name = "DPMSolverMultistepScheduler"

    def __init__(self, *a, algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True, **k):
        super().__init__(*a, algorithm_type=algorithm_type, solver_type=solver_type, lower_order_final=lower_order_final, **k)
        if algorithm_type not in ("dpmsolver", "dpmsolver++") or solver_type not in ("midpoint", "heun"):
            raise NotImplementedError(f"{algorithm_type}/{solver_type}")

    def convert_model_output(self, eps, t, sample):
        return self._x0_pred(eps, t, sample) if self.config.algorithm_type == "dpmsolver++" else eps

    def _coefs(self, ss: List[int], t: int):
        """Scalar coefficients (c_x, [c_m0, c_m1, c_m2]) of x_t = c_x*x + sum c_mi * m_i  (m0 = newest)."""
        pp = self.config.algorithm_type == "dpmsolver++"
        lam, al, sg = self.lambda_t.double(), self.alpha_t.double(), self.sigma_t.double()
        s0 = ss[-1]
        h = float(lam[t] - lam[s0])
        if pp:
            cx = float(sg[t] / sg[s0])
            e = math.exp(-h) - 1.0
            k0 = -float(al[t]) * e
        else:
            cx = float(al[t] / al[s0])
            e = math.exp(h) - 1.0
            k0 = -float(sg[t]) * e
        if len(ss) == 1:
            return cx, [k0]
        r0 = float(lam[s0] - lam[ss[-2]]) / h
        if len(ss) == 2:
            if self.config.solver_type == "midpoint":
                k1 = 0.5 * k0
            elif pp:
                k1 = float(al[t]) * (e / h + 1.0)
            else:
                k1 = -float(sg[t]) * (e / h - 1.0)
            return cx, [k0 + k1 / r0, -k1 / r0]
        r1 = float(lam[ss[-2]] - lam[ss[-3]]) / h
        if pp:
            k1 = float(al[t]) * (e / h + 1.0)
            k2 = -float(al[t]) * ((e + h) / h ** 2 - 0.5)
        else:
            k1 = -float(sg[t]) * (e / h - 1.0)
            k2 = -float(sg[t]) * ((e - h) / h ** 2 - 0.5)
        w, u = r0 / (r0 + r1), 1.0 / (r0 + r1)
        Pc, Qc = k1 * (1 + w) + k2 * u, -(k1 * w + k2 * u)
        return cx, [k0 + Pc / r0, -Pc / r0 + Qc / r1, -Qc / r1]

    def step(self, model_output, timestep, sample, return_dict=True, **_):
        t = int(timestep)
        i = self._step_index(t)
        n = len(self.timesteps)
        prev_t = 0 if i == n - 1 else int(self.timesteps[i + 1])
        lower_final = (i == n - 1) and self.config.lower_order_final and n < 15
        lower_second = (i == n - 2) and self.config.lower_order_final and n < 15
        m = self.convert_model_output(model_output, t, sample)
        order = self.config.solver_order
        for j in range(order - 1):
            self.model_outputs[j] = self.model_outputs[j + 1]
        self.model_outputs[-1] = m
        if order == 1 or self.lower_order_nums < 1 or lower_final:
            ss = [t]
        elif order == 2 or self.lower_order_nums < 2 or lower_second:
            ss = [int(self.timesteps[i - 1]), t]
        else:
            ss = [int(self.timesteps[i - 2]), int(self.timesteps[i - 1]), t]
        cx, cm = self._coefs(ss, prev_t)
        srcs = [sample.contiguous()] + [self.model_outputs[-1 - j].contiguous() for j in range(len(cm))]
        out = ops.lincomb(torch.empty_like(sample), srcs, [cx] + cm)
        if self.lower_order_nums < order:
            self.lower_order_nums += 1
        return SimpleNamespace(prev_sample=out)


class UniPCMultistepScheduler(_Multistep):
    """S4 -- [UPSTREAM] UniPCMultistepScheduler (bh1/bh2, predict_x0, UniC corrector + UniP predictor)."""
    _class_name = "UniPCMultistepScheduler"

    def __init__(self, *a, solver_type="bh2", predict_x0=True, lower_order_final=True, **k):
        super().__init__(*a, solver_type=solver_type, predict_x0=predict_x0, lower_order_final=lower_order_final, **k)
        self.timestep_list: List[Optional[int]] = [None] * self.config.solver_order
        self.last_sample = None
        self.this_order = 1

    def set_timesteps(self, num_inference_steps: int, device=None):
        super().set_timesteps(num_inference_steps)
        self.timestep_list = [None] * self.config.solver_order
        self.last_sample = None

    def convert_model_output(self, eps, t, sample):
        return self._x0_pred(eps, t, sample) if self.config.predict_x0 else eps

    def _rb(self, rks: torch.Tensor, order: int, hh: torch.Tensor):
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.config.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return torch.stack(R), torch.tensor(b), h_phi_1, B_h

    def _update(self, t: int, x: torch.Tensor, order: int, model_t: Optional[torch.Tensor]):
        """UniP (model_t None) or UniC (model_t = converted output at t): one vd_lincomb."""
        s0 = self.timestep_list[-1]
        h = self.lambda_t[t] - self.lambda_t[s0]
        rks = []
        hist = []                                   # older model outputs m_1.. (m0 is model_outputs[-1])
        for i in range(1, order):
            si = self.timestep_list[-(i + 1)]
            rks.append((self.lambda_t[si] - self.lambda_t[s0]) / h)
            hist.append(self.model_outputs[-(i + 1)])
        rks.append(1.0)
        rks = torch.tensor(rks)
        px0 = self.config.predict_x0
        hh = -h if px0 else h
        R, b, h_phi_1, B_h = self._rb(rks, order, hh)
        if model_t is None:                          # predictor
            if hist:
                rhos = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
            else:
                rhos = torch.zeros(0)
            rho_t = 0.0
        else:                                        # corrector
            rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
            rhos, rho_t = rhos_c[:-1], float(rhos_c[-1])
        lead = float(self.alpha_t[t]) if px0 else float(self.sigma_t[t])
        cx = float(self.sigma_t[t] / self.sigma_t[s0]) if px0 else float(self.alpha_t[t] / self.alpha_t[s0])
        bh = lead * float(B_h)
        c_m0 = -lead * float(h_phi_1)
        srcs, coefs = [x.contiguous()], [cx]
        for k_, mk in enumerate(hist):               # - lead*B_h*rho_k*(m_k - m0)/r_k
            ck = -bh * float(rhos[k_]) / float(rks[k_])
            srcs.append(mk)
            coefs.append(ck)
            c_m0 -= ck
        if model_t is not None:                      # - lead*B_h*rho_t*(model_t - m0)
            srcs.append(model_t)
            coefs.append(-bh * rho_t)
            c_m0 += bh * rho_t
        srcs.insert(1, self.model_outputs[-1])
        coefs.insert(1, c_m0)
        return ops.lincomb(torch.empty_like(x), srcs, coefs)

    def step(self, model_output, timestep, sample, return_dict=True, **_):
        t = int(timestep)
        i = self._step_index(t)
        n = len(self.timesteps)
        use_corr = i > 0 and self.last_sample is not None
        m = self.convert_model_output(model_output, t, sample)
        if use_corr:
            sample = self._update(t, self.last_sample, self.this_order, m)
        prev_t = 0 if i == n - 1 else int(self.timesteps[i + 1])
        order = self.config.solver_order
        for j in range(order - 1):
            self.model_outputs[j] = self.model_outputs[j + 1]
            self.timestep_list[j] = self.timestep_list[j + 1]
        self.model_outputs[-1] = m
        self.timestep_list[-1] = t
        this_order = min(order, n - i) if self.config.lower_order_final else order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        out = self._update(prev_t, sample, self.this_order, None)
        if self.lower_order_nums < order:
            self.lower_order_nums += 1
        return SimpleNamespace(prev_sample=out)


class PNDMScheduler(SchedulerBase):
    """[UPSTREAM] PNDMScheduler as the reference builds it (model.py:641-643): Runge-Kutta warm-up then PLMS.  Every step
    is ONE vd_lincomb: the RK / Adams-Bashforth weights of the stored eps history are folded into per-tensor scalars."""
    _class_name = "PNDMScheduler"

    def __init__(self, *a, skip_prk_steps=False, set_alpha_to_one=False, steps_offset=0, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, skip_prk_steps=skip_prk_steps, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **k)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.pndm_order = 4
        self.prk_timesteps = self.plms_timesteps = None
        self._reset()

    def _reset(self):
        self.counter, self.cur_sample, self.ets = 0, None, []
        self._rk: List[torch.Tensor] = []            # eps of the current Runge-Kutta stage sequence

    def set_timesteps(self, num_inference_steps: int, device=None):
        T, n = self.config.num_train_timesteps, num_inference_steps
        self.num_inference_steps = n
        base = (np.arange(0, n) * (T // n)).round() + self.config.steps_offset
        if self.config.skip_prk_steps:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        else:
            prk = np.array(base[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, T // n // 2]), self.pndm_order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = base[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self._reset()

    def _prev(self, sample, t, prev_t, outs: List[torch.Tensor], weights: List[float]):
        """x_prev = c_s*sample - c_m * sum_j w_j outs_j  (upstream _get_prev_sample with model_output = sum_j w_j outs_j)."""
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        c_s = (a_prev / a_t) ** 0.5
        c_m = -float((a_prev - a_t) / (a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5))
        return ops.lincomb(torch.empty_like(sample), [sample.contiguous()] + [o.contiguous() for o in outs],
                           [float(c_s)] + [c_m * w for w in weights])

    def step(self, model_output, timestep, sample, return_dict=True, **_):
        t = int(timestep)
        T, n = self.config.num_train_timesteps, self.num_inference_steps
        if self.counter < len(self.prk_timesteps) and not self.config.skip_prk_steps:
            c = self.counter
            prev_t = t - (0 if c % 2 else T // n // 2)
            t0 = int(self.prk_timesteps[c // 4 * 4])
            stage = c % 4
            if stage == 0:
                self._rk = [model_output]
                self.ets.append(model_output)
                self.cur_sample = sample
            else:
                self._rk.append(model_output)
            if stage == 3:
                outs, w = self._rk, [1 / 6, 1 / 3, 1 / 3, 1 / 6]
            else:
                outs, w = [model_output], [1.0]
            out = self._prev(self.cur_sample if self.cur_sample is not None else sample, t0, prev_t, outs, w)
            self.counter += 1
            return SimpleNamespace(prev_sample=out)
        prev_t = t - T // n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t, t = t, t + T // n
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            outs, w = [model_output], [1.0]
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            outs, w = [model_output, e[-1]], [0.5, 0.5]
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            outs, w = [e[-1], e[-2]], [1.5, -0.5]
        elif len(e) == 3:
            outs, w = [e[-1], e[-2], e[-3]], [23 / 12, -16 / 12, 5 / 12]
        else:
            outs, w = [e[-1], e[-2], e[-3], e[-4]], [55 / 24, -59 / 24, 37 / 24, -9 / 24]
        out = self._prev(sample, t, prev_t, outs, w)
        self.counter += 1
        return SimpleNamespace(prev_sample=out)


class DEISMultistepScheduler(_Multistep):
    """[UPSTREAM] DEISMultistepScheduler (reference model.py:644-646: defaults, order 2, 'deis' / 'logrho')."""
    _class_name = "DEISMultistepScheduler"

    def __init__(self, *a, algorithm_type="deis", solver_type="logrho", lower_order_final=True, **k):
        super().__init__(*a, algorithm_type=algorithm_type, solver_type=solver_type, lower_order_final=lower_order_final, **k)
        if algorithm_type != "deis" or solver_type != "logrho":
            raise NotImplementedError(f"{algorithm_type}/{solver_type}")

    def convert_model_output(self, eps, t, sample):
        """upstream: eps -> x0 -> eps round trip (the x0 leg is the thresholding hook)."""
        x0 = self._x0_pred(eps, t, sample)
        return ops.lincomb(torch.empty_like(sample), [sample.contiguous(), x0],
                           [float(1.0 / self.sigma_t[t]), -float(self.alpha_t[t] / self.sigma_t[t])])

    def _coefs(self, ss: List[int], t: int):
        al, sg, lam = self.alpha_t.double(), self.sigma_t.double(), self.lambda_t.double()
        s0 = ss[-1]
        at = float(al[t])
        if len(ss) == 1:
            h = float(lam[t] - lam[s0])
            return float(al[t] / al[s0]), [-float(sg[t]) * (math.exp(h) - 1.0)]
        rho = lambda u: float(sg[u] / al[u])
        lg = math.log
        if len(ss) == 2:
            rt, r0, r1 = rho(t), rho(s0), rho(ss[-2])
            ind = lambda q, b, c: q * (-lg(c) + lg(q) - 1) / (lg(b) - lg(c))
            c1 = ind(rt, r0, r1) - ind(r0, r0, r1)
            c2 = ind(rt, r1, r0) - ind(r0, r1, r0)
            return at / float(al[s0]), [at * c1, at * c2]
        rt, r0, r1, r2 = rho(t), rho(s0), rho(ss[-2]), rho(ss[-3])

        def ind(q, b, c, d):
            lq, lb, lc, ld = lg(q), lg(b), lg(c), lg(d)
            return q * (lc * (ld - lq + 1) - ld * lq + ld + lq ** 2 - 2 * lq + 2) / ((lb - lc) * (lb - ld))

        c1 = ind(rt, r0, r1, r2) - ind(r0, r0, r1, r2)
        c2 = ind(rt, r1, r2, r0) - ind(r0, r1, r2, r0)
        c3 = ind(rt, r2, r0, r1) - ind(r0, r2, r0, r1)
        return at / float(al[s0]), [at * c1, at * c2, at * c3]

    step = DPMSolverMultistepScheduler.step


class _SigmaSpace(SchedulerBase):
    """k-diffusion style samplers on the VP network: sigma = sqrt((1-abar)/abar); the state is x_vp*sqrt(sigma^2+1), so the
    pipeline starts from init*init_noise_sigma and feeds the UNet scale_model_input(x, t) with FLOAT timesteps."""

    def __init__(self, *a, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, **k)
        self.set_timesteps(self.config.num_train_timesteps)

    def _interp_sigmas(self, n):
        T = self.config.num_train_timesteps
        ts = np.linspace(0, T - 1, n, dtype=float)[::-1].copy()
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        return ts, np.concatenate([sig, [0.0]]).astype(np.float32)

    def _scaled(self, sample, sigma):
        return ops.lincomb(torch.empty_like(sample), [sample.contiguous()], [float(1.0 / ((sigma ** 2 + 1) ** 0.5))])


class HeunDiscreteScheduler(_SigmaSpace):
    """[UPSTREAM] HeunDiscreteScheduler (reference model.py:647-649): 2n-1 UNet calls, inner timesteps repeated."""
    _class_name = "HeunDiscreteScheduler"

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts, sig = self._interp_sigmas(num_inference_steps)
        sig = torch.from_numpy(sig)
        self.sigmas = torch.cat([sig[:1], sig[1:-1].repeat_interleave(2), sig[-1:]])
        self.init_noise_sigma = self.sigmas.max()
        ts = torch.from_numpy(ts)
        self.timesteps = torch.cat([ts[:1], ts[1:].repeat_interleave(2)])
        self._eps1 = self._x1 = self._s = None

    @property
    def state_in_first_order(self):
        return self._s is None

    def index_for_timestep(self, timestep):
        idx = (self.timesteps == timestep).nonzero()
        return int(idx[-1 if self.state_in_first_order else 0].item())

    def scale_model_input(self, sample, timestep=None):
        return self._scaled(sample, self.sigmas[self.index_for_timestep(timestep)])

    def step(self, model_output, timestep, sample, return_dict=True, **_):
        i = self.index_for_timestep(timestep)
        if self.state_in_first_order:
            sigma, sigma_next = self.sigmas[i], self.sigmas[i + 1]
            # derivative = (x - (x - sigma*eps))/sigma = eps (up to rounding);  x' = x + eps*dt
            dt = float(sigma_next - sigma)
            out = ops.lincomb(torch.empty_like(sample), [sample.contiguous(), model_output.contiguous()], [1.0, dt])
            self._eps1, self._x1, self._s = model_output, sample, dt
        else:
            dt = self._s
            out = ops.lincomb(torch.empty_like(sample), [self._x1.contiguous(), self._eps1.contiguous(), model_output.contiguous()],
                              [1.0, 0.5 * dt, 0.5 * dt])
            self._eps1 = self._x1 = self._s = None
        return SimpleNamespace(prev_sample=out)


class LMSDiscreteScheduler(_SigmaSpace):
    """[UPSTREAM] LMSDiscreteScheduler (reference model.py:650-652): order-4 linear multistep, Lagrange-basis integrals by
    scipy.integrate.quad(epsrel=1e-4) like upstream."""
    _class_name = "LMSDiscreteScheduler"

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts, sig = self._interp_sigmas(num_inference_steps)
        self.sigmas = torch.from_numpy(sig)
        self.init_noise_sigma = self.sigmas.max()
        self.timesteps = torch.from_numpy(ts)
        self.derivatives: List[torch.Tensor] = []

    def _index(self, timestep):
        return int((self.timesteps == timestep).nonzero().item())

    def scale_model_input(self, sample, timestep=None):
        return self._scaled(sample, self.sigmas[self._index(timestep)])

    def get_lms_coefficient(self, order, t, current_order):
        from scipy import integrate

        def lms_derivative(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
            return prod

        return integrate.quad(lms_derivative, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]

    def step(self, model_output, timestep, sample, order: int = 4, return_dict=True, **_):
        i = self._index(timestep)
        self.derivatives.append(model_output)            # (x - (x - sigma*eps))/sigma == eps up to rounding
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [float(self.get_lms_coefficient(order, i, c)) for c in range(order)]
        ders = list(reversed(self.derivatives))[:order]
        out = ops.lincomb(torch.empty_like(sample), [sample.contiguous()] + [d.contiguous() for d in ders], [1.0] + coeffs[:len(ders)])
        return SimpleNamespace(prev_sample=out)


class KarrasVeScheduler:
    """[UPSTREAM] KarrasVeScheduler (reference model.py:685-693: sigma_min 0.01, sigma_max 380, s_churn 80 / 100 / 0).
    `schedule` holds sigma_max^2 (sigma_min^2/sigma_max^2)^(i/(n-1)) and is used as sigma, exactly as upstream does."""
    _class_name = "KarrasVeScheduler"
    order = 2

    def __init__(self, sigma_min=0.02, sigma_max=100.0, s_noise=1.007, s_churn=80.0, s_min=0.05, s_max=50.0,
                 num_train_timesteps=None, **extra):
        self.config = _Config(sigma_min=sigma_min, sigma_max=sigma_max, s_noise=s_noise, s_churn=s_churn, s_min=s_min, s_max=s_max,
                              num_train_timesteps=num_train_timesteps, clip_sample=False)
        self.init_noise_sigma = sigma_max
        self.num_inference_steps = None
        self.timesteps = self.schedule = None
        self.device_rng_seed: Optional[int] = None
        self._rng_offset = 0

    def scheduler_config(self) -> dict:
        d = {k: v for k, v in vars(self.config).items() if k != "clip_sample"}
        d["_class_name"] = self._class_name
        return d

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy())
        c = self.config
        self.schedule = torch.tensor([c.sigma_max ** 2 * (c.sigma_min ** 2 / c.sigma_max ** 2) ** (i / (n - 1)) for i in self.timesteps],
                                     dtype=torch.float32)

    def add_noise_to_input(self, sample, sigma, generator=None, noise=None):
        c = self.config
        gamma = min(c.s_churn / self.num_inference_steps, 2 ** 0.5 - 1) if c.s_min <= sigma <= c.s_max else 0
        sigma_hat = sigma + gamma * sigma
        cz = float((sigma_hat ** 2 - sigma ** 2) ** 0.5) * c.s_noise
        if noise is None:
            if self.device_rng_seed is not None:
                noise = ops.randn(torch.empty_like(sample), int(self.device_rng_seed), self._rng_offset)
                self._rng_offset += (sample.numel() + 3) // 4
            else:
                noise = _noise_like(sample, generator)
        out = ops.lincomb(torch.empty_like(sample), [sample.contiguous(), noise.contiguous()], [1.0, cz])
        return out, sigma_hat

    def step(self, model_output, sigma_hat, sigma_prev, sample_hat, return_dict=True):
        """derivative = (x_hat - (x_hat + sigma_hat*mo))/sigma_hat = -mo ; x_prev = x_hat + (sigma_prev - sigma_hat)*derivative."""
        mo = model_output.contiguous()
        derivative = ops.lincomb(torch.empty_like(mo), [mo], [-1.0])
        out = ops.lincomb(torch.empty_like(sample_hat), [sample_hat.contiguous(), derivative], [1.0, float(sigma_prev - sigma_hat)])
        return SimpleNamespace(prev_sample=out, derivative=derivative)

    def step_correct(self, model_output, sigma_hat, sigma_prev, sample_hat, sample_prev, derivative, return_dict=True):
        d = 0.5 * float(sigma_prev - sigma_hat)
        out = ops.lincomb(torch.empty_like(sample_hat), [sample_hat.contiguous(), derivative.contiguous(), model_output.contiguous()],
                          [1.0, d, -d])
        return SimpleNamespace(prev_sample=out, derivative=derivative)


class ScoreSdeVeScheduler:
    """S5 -- [UPSTREAM] ScoreSdeVeScheduler (predictor-corrector VE-SDE sampler; reference model.py:672-684).
    Per-sample norms are one reduction kernel, both state updates are one ``vd_lincomb`` each."""
    _class_name = "ScoreSdeVeScheduler"
    order = 1

    def __init__(self, num_train_timesteps=2000, snr=0.15, sigma_min=0.01, sigma_max=1348.0, sampling_eps=1e-5, correct_steps=1, **extra):
        extra.setdefault("clip_sample", False)           # (a saved scheduler_config.json carries it back in)
        self.config = _Config(num_train_timesteps=num_train_timesteps, snr=snr, sigma_min=sigma_min, sigma_max=sigma_max,
                              sampling_eps=sampling_eps, correct_steps=correct_steps, **extra)
        self.init_noise_sigma = sigma_max
        self.timesteps = None
        self.device_rng_seed: Optional[int] = None
        self._rng_offset = 0
        self.set_sigmas(num_train_timesteps)

    def scheduler_config(self) -> dict:
        d = {k: v for k, v in vars(self.config).items()}
        d["_class_name"] = self._class_name
        return d

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, sampling_eps: float = None, device=None):
        eps = sampling_eps if sampling_eps is not None else self.config.sampling_eps
        self.timesteps = torch.linspace(1, eps, num_inference_steps)

    def set_sigmas(self, num_inference_steps: int, sigma_min=None, sigma_max=None, sampling_eps=None):
        smin = self.config.sigma_min if sigma_min is None else sigma_min
        smax = self.config.sigma_max if sigma_max is None else sigma_max
        if self.timesteps is None:
            self.set_timesteps(num_inference_steps, sampling_eps)
        self.discrete_sigmas = torch.exp(torch.linspace(math.log(smin), math.log(smax), num_inference_steps))
        self.sigmas = torch.tensor([smin * (smax / smin) ** t for t in self.timesteps])

    def _z(self, x, generator, noise):
        if noise is not None:
            return noise
        if self.device_rng_seed is not None:
            z = torch.empty_like(x)
            ops.randn(z, int(self.device_rng_seed), self._rng_offset)
            self._rng_offset += (x.numel() + 3) // 4
            return z
        return _noise_like(x, generator)

    def step_correct(self, model_output, sample, generator=None, noise=None, return_dict=True):
        z = self._z(sample, generator, noise).contiguous()
        B = sample.shape[0]
        norms = torch.empty(2 * B, device=sample.device, dtype=torch.float32)
        ops.batch_l2norm(model_output.contiguous(), norms[:B])
        ops.batch_l2norm(z, norms[B:])
        nrm = norms.cpu()
        gnorm, znorm = nrm[:B].mean(), nrm[B:].mean()
        step = (self.config.snr * znorm / gnorm) ** 2 * 2
        mean = ops.lincomb(torch.empty_like(sample), [sample.contiguous(), model_output.contiguous()], [1.0, float(step)])
        prev = ops.lincomb(torch.empty_like(sample), [mean, z], [1.0, float((step * 2) ** 0.5)])
        return SimpleNamespace(prev_sample=prev, prev_sample_mean=mean)

    def step_pred(self, model_output, timestep, sample, generator=None, noise=None, return_dict=True):
        t = float(timestep)
        idx = int(torch.tensor(t * (len(self.timesteps) - 1)).long())           # (timestep * (N-1)).long()
        sigma = self.discrete_sigmas[idx]
        adj = torch.zeros(()) if idx == 0 else self.discrete_sigmas[idx - 1]
        diffusion = (sigma ** 2 - adj ** 2) ** 0.5
        z = self._z(sample, generator, noise).contiguous()
        mean = ops.lincomb(torch.empty_like(sample), [sample.contiguous(), model_output.contiguous()], [1.0, float(diffusion ** 2)])
        prev = ops.lincomb(torch.empty_like(sample), [mean, z], [1.0, float(diffusion)])
        return SimpleNamespace(prev_sample=prev, prev_sample_mean=mean)


def get_cosine_schedule_with_warmup_lambda(num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
    """[UPSTREAM] diffusers.optimization.get_cosine_schedule_with_warmup's lr_lambda (VillanDiffusion.py:446-450)."""
    def lr_lambda(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
    return lr_lambda


SCHEDULER_CLASSES = {c._class_name: c for c in (DDPMScheduler, DDIMScheduler, DPMSolverMultistepScheduler,
                                                UniPCMultistepScheduler, ScoreSdeVeScheduler, PNDMScheduler,
                                                DEISMultistepScheduler, HeunDiscreteScheduler, LMSDiscreteScheduler,
                                                KarrasVeScheduler)}
