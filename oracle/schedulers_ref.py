"""CPU oracle: the reverse-diffusion samplers the reference selects in model.py:599-776.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the scheduler
arithmetic is in the un-vendored diffusers fork (requirement.txt:37); this file
restates the published upstream algorithms (diffusers ~0.16: DDPMScheduler,
DDIMScheduler, DPMSolverMultistepScheduler, UniPCMultistepScheduler,
ScoreSdeVeScheduler, PNDMScheduler, DEISMultistepScheduler, HeunDiscreteScheduler,
LMSDiscreteScheduler, KarrasVeScheduler) as summarised in SURVEY.md §8a rows S1-S5, with every
coefficient computed by the same fp32 torch op sequence upstream uses.
Checked by analytic identities in tests/test_schedulers_oracle.py.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch


def make_betas(num_train_timesteps: int, beta_start: float, beta_end: float, beta_schedule: str) -> torch.Tensor:
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


def _randn(shape, generator, device, dtype):
    """[UPSTREAM] randn_tensor: a CPU generator draws on the CPU and the result is moved."""
    gdev = generator.device if generator is not None else device
    return torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)


class _VPBase:
    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, clip_sample_range=1.0):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule,
                                      clip_sample=clip_sample, clip_sample_range=clip_sample_range)
        self.betas = make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def add_noise(self, x0, noise, timesteps):
        ac = self.alphas_cumprod.to(device=x0.device, dtype=x0.dtype)
        t = timesteps.to(x0.device)
        sa = (ac[t] ** 0.5).flatten()
        sb = ((1 - ac[t]) ** 0.5).flatten()
        while sa.dim() < x0.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * x0 + sb * noise


class DDPMSchedulerRef(_VPBase):
    """S1.  [UPSTREAM diffusers ~0.16 DDPMScheduler._get_variance / step] epsilon prediction; variance_type fixed_small (posterior
    variance clamped at 1e-20), fixed_small_log (exp(0.5 * log(clamp))), fixed_large (beta_t; what google/ddpm-cifar10-32 ships).
    fixed_large_log takes sqrt(log(beta)) upstream (NaN), learned / learned_range need a 2C-channel network: rejected."""
    VARIANCE_TYPES = ("fixed_small", "fixed_small_log", "fixed_large")

    def __init__(self, *a, variance_type="fixed_small", **k):
        super().__init__(*a, **k)
        if variance_type not in self.VARIANCE_TYPES:
            raise NotImplementedError(f"variance_type {variance_type}")
        self.config.variance_type = variance_type

    def noise_scale(self, a_t, a_prev, cur_beta):
        """The factor of z in the reverse step (std, not variance)."""
        vt = self.config.variance_type
        if vt == "fixed_large":
            return cur_beta ** 0.5
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
        if vt == "fixed_small_log":
            return torch.exp(0.5 * torch.log(var))
        return var ** 0.5

    def set_timesteps(self, n: int):
        T = self.config.num_train_timesteps
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        t = int(timestep)
        T = self.config.num_train_timesteps
        n = self.num_inference_steps if self.num_inference_steps else T
        prev_t = t - T // n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        c_x0 = (a_prev ** 0.5 * cur_beta) / b_t
        c_xt = cur_alpha ** 0.5 * b_prev / b_t
        prev = c_x0 * x0 + c_xt * sample
        if t > 0:
            z = noise if noise is not None else _randn(model_output.shape, generator, model_output.device, model_output.dtype)
            prev = prev + self.noise_scale(a_t, a_prev, cur_beta) * z
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DDIMSchedulerRef(_VPBase):
    """S2."""

    def __init__(self, *a, set_alpha_to_one=True, steps_offset=0, **k):
        super().__init__(*a, **k)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset

    def set_timesteps(self, n: int):
        T = self.config.num_train_timesteps
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, noise=None):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            z = noise if noise is not None else _randn(model_output.shape, generator, model_output.device, model_output.dtype)
            prev = prev + std * z
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class _MultistepBase(_VPBase):
    def __init__(self, *a, solver_order=2, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, **k)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.config.solver_order = solver_order
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.lower_order_nums = 0

    def set_timesteps(self, n: int):
        T = self.config.num_train_timesteps
        self.num_inference_steps = n
        ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0

    def _step_index(self, timestep) -> int:
        idx = (self.timesteps == int(timestep)).nonzero()
        return len(self.timesteps) - 1 if len(idx) == 0 else int(idx[0].item())


class DPMSolverMultistepSchedulerRef(_MultistepBase):
    """S3.  algorithm_type in {dpmsolver, dpmsolver++}, solver_type midpoint, lower_order_final."""

    def __init__(self, *a, algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True, **k):
        super().__init__(*a, **k)
        self.config.algorithm_type, self.config.solver_type = algorithm_type, solver_type
        self.config.lower_order_final = lower_order_final

    def convert_model_output(self, eps, t, sample):
        if self.config.algorithm_type == "dpmsolver++":
            return (sample - self.sigma_t[t] * eps) / self.alpha_t[t]
        return eps

    def _first(self, m, s, t, x):
        lam_t, lam_s = self.lambda_t[t], self.lambda_t[s]
        h = lam_t - lam_s
        if self.config.algorithm_type == "dpmsolver++":
            return (self.sigma_t[t] / self.sigma_t[s]) * x - (self.alpha_t[t] * (torch.exp(-h) - 1.0)) * m
        return (self.alpha_t[t] / self.alpha_t[s]) * x - (self.sigma_t[t] * (torch.exp(h) - 1.0)) * m

    def _second(self, ms, ss, t, x):
        s0, s1 = ss[-1], ss[-2]
        m0, m1 = ms[-1], ms[-2]
        lam_t, lam_s0, lam_s1 = self.lambda_t[t], self.lambda_t[s0], self.lambda_t[s1]
        h, h0 = lam_t - lam_s0, lam_s0 - lam_s1
        r0 = h0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        if self.config.algorithm_type == "dpmsolver++":
            a = self.alpha_t[t] * (torch.exp(-h) - 1.0)
            if self.config.solver_type == "midpoint":
                return (self.sigma_t[t] / self.sigma_t[s0]) * x - a * D0 - 0.5 * a * D1
            return (self.sigma_t[t] / self.sigma_t[s0]) * x - a * D0 + (self.alpha_t[t] * ((torch.exp(-h) - 1.0) / h + 1.0)) * D1
        a = self.sigma_t[t] * (torch.exp(h) - 1.0)
        if self.config.solver_type == "midpoint":
            return (self.alpha_t[t] / self.alpha_t[s0]) * x - a * D0 - 0.5 * a * D1
        return (self.alpha_t[t] / self.alpha_t[s0]) * x - a * D0 - (self.sigma_t[t] * ((torch.exp(h) - 1.0) / h - 1.0)) * D1

    def _third(self, ms, ss, t, x):
        s0, s1, s2 = ss[-1], ss[-2], ss[-3]
        m0, m1, m2 = ms[-1], ms[-2], ms[-3]
        lam = self.lambda_t
        h, h0, h1 = lam[t] - lam[s0], lam[s0] - lam[s1], lam[s1] - lam[s2]
        r0, r1 = h0 / h, h1 / h
        D0 = m0
        D1_0, D1_1 = (1.0 / r0) * (m0 - m1), (1.0 / r1) * (m1 - m2)
        D1 = D1_0 + (r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = (1.0 / (r0 + r1)) * (D1_0 - D1_1)
        if self.config.algorithm_type == "dpmsolver++":
            at = self.alpha_t[t]
            return ((self.sigma_t[t] / self.sigma_t[s0]) * x - (at * (torch.exp(-h) - 1.0)) * D0
                    + (at * ((torch.exp(-h) - 1.0) / h + 1.0)) * D1
                    - (at * ((torch.exp(-h) - 1.0 + h) / h ** 2 - 0.5)) * D2)
        st = self.sigma_t[t]
        return ((self.alpha_t[t] / self.alpha_t[s0]) * x - (st * (torch.exp(h) - 1.0)) * D0
                - (st * ((torch.exp(h) - 1.0) / h - 1.0)) * D1
                - (st * ((torch.exp(h) - 1.0 - h) / h ** 2 - 0.5)) * D2)

    def step(self, model_output, timestep, sample, **_):
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
            prev = self._first(m, t, prev_t, sample)
        elif order == 2 or self.lower_order_nums < 2 or lower_second:
            prev = self._second(self.model_outputs, [int(self.timesteps[i - 1]), t], prev_t, sample)
        else:
            prev = self._third(self.model_outputs, [int(self.timesteps[i - 2]), int(self.timesteps[i - 1]), t], prev_t, sample)
        if self.lower_order_nums < order:
            self.lower_order_nums += 1
        return SimpleNamespace(prev_sample=prev)


class UniPCMultistepSchedulerRef(_MultistepBase):
    """S4.  bh2, predict_x0, solver_order 2, lower_order_final."""

    def __init__(self, *a, solver_type="bh2", predict_x0=True, lower_order_final=True, **k):
        super().__init__(*a, **k)
        self.config.solver_type, self.config.predict_x0 = solver_type, predict_x0
        self.config.lower_order_final = lower_order_final
        self.timestep_list = [None] * self.config.solver_order
        self.last_sample = None
        self.this_order = 1

    def set_timesteps(self, n: int):
        super().set_timesteps(n)
        self.timestep_list = [None] * self.config.solver_order
        self.last_sample = None

    def convert_model_output(self, eps, t, sample):
        if self.config.predict_x0:
            return (sample - self.sigma_t[t] * eps) / self.alpha_t[t]
        return eps

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

    def _history(self, order, s0, h):
        rks, D1s = [], []
        m0 = self.model_outputs[-1]
        for i in range(1, order):
            si, mi = self.timestep_list[-(i + 1)], self.model_outputs[-(i + 1)]
            rk = (self.lambda_t[si] - self.lambda_t[s0]) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        return torch.tensor(rks), (torch.stack(D1s, dim=1) if D1s else None)

    def _uni_p(self, prev_t, x, order):
        s0, t = self.timestep_list[-1], prev_t
        m0 = self.model_outputs[-1]
        h = self.lambda_t[t] - self.lambda_t[s0]
        rks, D1s = self._history(order, s0, h)
        hh = -h if self.config.predict_x0 else h
        R, b, h_phi_1, B_h = self._rb(rks, order, hh)
        if D1s is not None:
            rhos = torch.tensor([0.5], dtype=x.dtype) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
            res = torch.einsum("k,bkchw->bchw", rhos, D1s)
        else:
            res = 0
        if self.config.predict_x0:
            return self.sigma_t[t] / self.sigma_t[s0] * x - self.alpha_t[t] * h_phi_1 * m0 - self.alpha_t[t] * B_h * res
        return self.alpha_t[t] / self.alpha_t[s0] * x - self.sigma_t[t] * h_phi_1 * m0 - self.sigma_t[t] * B_h * res

    def _uni_c(self, model_t, this_t, last_sample, order):
        s0, t = self.timestep_list[-1], this_t
        m0 = self.model_outputs[-1]
        x = last_sample
        h = self.lambda_t[t] - self.lambda_t[s0]
        rks, D1s = self._history(order, s0, h)
        hh = -h if self.config.predict_x0 else h
        R, b, h_phi_1, B_h = self._rb(rks, order, hh)
        rhos = torch.tensor([0.5], dtype=x.dtype) if order == 1 else torch.linalg.solve(R, b)
        corr = torch.einsum("k,bkchw->bchw", rhos[:-1], D1s) if D1s is not None else 0
        D1_t = model_t - m0
        if self.config.predict_x0:
            return (self.sigma_t[t] / self.sigma_t[s0] * x - self.alpha_t[t] * h_phi_1 * m0
                    - self.alpha_t[t] * B_h * (corr + rhos[-1] * D1_t))
        return (self.alpha_t[t] / self.alpha_t[s0] * x - self.sigma_t[t] * h_phi_1 * m0
                - self.sigma_t[t] * B_h * (corr + rhos[-1] * D1_t))

    def step(self, model_output, timestep, sample, **_):
        t = int(timestep)
        i = self._step_index(t)
        n = len(self.timesteps)
        use_corr = i > 0 and self.last_sample is not None
        m = self.convert_model_output(model_output, t, sample)
        if use_corr:
            sample = self._uni_c(m, t, self.last_sample, self.this_order)
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
        prev = self._uni_p(prev_t, sample, self.this_order)
        if self.lower_order_nums < order:
            self.lower_order_nums += 1
        return SimpleNamespace(prev_sample=prev)


class PNDMSchedulerRef(_VPBase):
    """[UPSTREAM] PNDMScheduler (reference model.py:641-643 builds it with defaults: skip_prk_steps=False, set_alpha_to_one=False,
    steps_offset=0, epsilon prediction): 4th-order Runge-Kutta warm-up (12 UNet calls) then linear multistep (PLMS)."""

    def __init__(self, *a, skip_prk_steps=False, set_alpha_to_one=False, steps_offset=0, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, **k)
        self.config.skip_prk_steps, self.config.set_alpha_to_one, self.config.steps_offset = skip_prk_steps, set_alpha_to_one, steps_offset
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.pndm_order = 4
        self.cur_model_output, self.counter, self.cur_sample, self.ets = 0, 0, None, []
        self.prk_timesteps = self.plms_timesteps = None

    def set_timesteps(self, n: int):
        T = self.config.num_train_timesteps
        self.num_inference_steps = n
        step_ratio = T // n
        self._timesteps = (np.arange(0, n) * step_ratio).round() + self.config.steps_offset
        if self.config.skip_prk_steps:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = np.concatenate([self._timesteps[:-1], self._timesteps[-2:-1], self._timesteps[-1:]])[::-1].copy()
        else:
            prk = np.array(self._timesteps[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, T // n // 2]), self.pndm_order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = self._timesteps[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.ets, self.counter, self.cur_model_output, self.cur_sample = [], 0, 0, None

    def step(self, model_output, timestep, sample, **_):
        if self.counter < len(self.prk_timesteps) and not self.config.skip_prk_steps:
            return self.step_prk(model_output, int(timestep), sample)
        return self.step_plms(model_output, int(timestep), sample)

    def step_prk(self, model_output, timestep, sample):
        T, n = self.config.num_train_timesteps, self.num_inference_steps
        diff_to_prev = 0 if self.counter % 2 else T // n // 2
        prev_timestep = timestep - diff_to_prev
        timestep = int(self.prk_timesteps[self.counter // 4 * 4])
        if self.counter % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 6 * model_output
            self.ets.append(model_output)
            self.cur_sample = sample
        elif (self.counter - 1) % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
        elif (self.counter - 2) % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
        elif (self.counter - 3) % 4 == 0:
            model_output = self.cur_model_output + 1 / 6 * model_output
            self.cur_model_output = 0
        cur_sample = self.cur_sample if self.cur_sample is not None else sample
        prev = self._get_prev_sample(cur_sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return SimpleNamespace(prev_sample=prev)

    def step_plms(self, model_output, timestep, sample):
        T, n = self.config.num_train_timesteps, self.num_inference_steps
        prev_timestep = timestep - T // n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + T // n
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = (model_output + e[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(e) == 2:
            model_output = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            model_output = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        prev = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return SimpleNamespace(prev_sample=prev)

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) * model_output / denom


class DEISMultistepSchedulerRef(_MultistepBase):
    """[UPSTREAM] DEISMultistepScheduler (reference model.py:644-646, defaults: solver_order=2, algorithm_type 'deis',
    solver_type 'logrho', lower_order_final=True): exponential integrator with a log-rho polynomial fit of eps."""

    def __init__(self, *a, lower_order_final=True, **k):
        super().__init__(*a, **k)
        self.config.lower_order_final = lower_order_final

    def convert_model_output(self, eps, t, sample):
        a, sg = self.alpha_t[t], self.sigma_t[t]
        x0 = (sample - sg * eps) / a
        return (sample - a * x0) / sg            # upstream round-trips eps through x0 (thresholding hook)

    def _first(self, m, s, t, x):
        h = self.lambda_t[t] - self.lambda_t[s]
        return (self.alpha_t[t] / self.alpha_t[s]) * x - (self.sigma_t[t] * (torch.exp(h) - 1.0)) * m

    def _second(self, ms, ss, t, x):
        s0, s1 = ss[-1], ss[-2]
        m0, m1 = ms[-1], ms[-2]
        at, a0, a1 = self.alpha_t[t], self.alpha_t[s0], self.alpha_t[s1]
        rt, r0, r1 = self.sigma_t[t] / at, self.sigma_t[s0] / a0, self.sigma_t[s1] / a1

        def ind(t_, b, c):
            return t_ * (-torch.log(c) + torch.log(t_) - 1) / (torch.log(b) - torch.log(c))

        c1 = ind(rt, r0, r1) - ind(r0, r0, r1)
        c2 = ind(rt, r1, r0) - ind(r0, r1, r0)
        return at * (x / a0 + c1 * m0 + c2 * m1)

    def _third(self, ms, ss, t, x):
        s0, s1, s2 = ss[-1], ss[-2], ss[-3]
        m0, m1, m2 = ms[-1], ms[-2], ms[-3]
        at, a0, a1, a2 = self.alpha_t[t], self.alpha_t[s0], self.alpha_t[s1], self.alpha_t[s2]
        rt, r0, r1, r2 = self.sigma_t[t] / at, self.sigma_t[s0] / a0, self.sigma_t[s1] / a1, self.sigma_t[s2] / a2

        def ind(t_, b, c, d):
            lt, lb, lc, ld = torch.log(t_), torch.log(b), torch.log(c), torch.log(d)
            num = t_ * (lc * (ld - lt + 1) - ld * lt + ld + lt ** 2 - 2 * lt + 2)
            return num / ((lb - lc) * (lb - ld))

        c1 = ind(rt, r0, r1, r2) - ind(r0, r0, r1, r2)
        c2 = ind(rt, r1, r2, r0) - ind(r0, r1, r2, r0)
        c3 = ind(rt, r2, r0, r1) - ind(r0, r2, r0, r1)
        return at * (x / a0 + c1 * m0 + c2 * m1 + c3 * m2)

    step = DPMSolverMultistepSchedulerRef.step


class _SigmaBase(_VPBase):
    """k-diffusion style samplers on the VP model: sigma = sqrt((1-abar)/abar), state x_sigma = x_vp * sqrt(sigma^2+1)."""

    def _interp_sigmas(self, n):
        T = self.config.num_train_timesteps
        ts = np.linspace(0, T - 1, n, dtype=float)[::-1].copy()
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        return ts, np.concatenate([sig, [0.0]]).astype(np.float32)


class HeunDiscreteSchedulerRef(_SigmaBase):
    """[UPSTREAM] HeunDiscreteScheduler (reference model.py:647-649): Heun's 2nd-order method in sigma space (Karras
    et al. 2022, Alg. 1 with gamma=0); 2n-1 UNet calls; timesteps are floats, every inner one repeated."""

    def __init__(self, *a, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, **k)
        self.set_timesteps(self.config.num_train_timesteps)

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ts, sig = self._interp_sigmas(n)
        sig = torch.from_numpy(sig)
        self.sigmas = torch.cat([sig[:1], sig[1:-1].repeat_interleave(2), sig[-1:]])
        self.init_noise_sigma = self.sigmas.max()
        ts = torch.from_numpy(ts)
        self.timesteps = torch.cat([ts[:1], ts[1:].repeat_interleave(2)])
        self.prev_derivative = self.dt = self.sample = None

    @property
    def state_in_first_order(self):
        return self.dt is None

    def index_for_timestep(self, timestep):
        idx = (self.timesteps == timestep).nonzero()
        return int(idx[-1 if self.state_in_first_order else 0].item())

    def scale_model_input(self, sample, timestep):
        sigma = self.sigmas[self.index_for_timestep(timestep)]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, **_):
        i = self.index_for_timestep(timestep)
        if self.state_in_first_order:
            sigma, sigma_next = self.sigmas[i], self.sigmas[i + 1]
        else:
            sigma, sigma_next = self.sigmas[i - 1], self.sigmas[i]
        sigma_hat = sigma * (0 + 1)
        sigma_input = sigma_hat if self.state_in_first_order else sigma_next
        pred_original = sample - sigma_input * model_output
        if self.state_in_first_order:
            derivative = (sample - pred_original) / sigma_hat
            dt = sigma_next - sigma_hat
            self.prev_derivative, self.dt, self.sample = derivative, dt, sample
        else:
            derivative = (sample - pred_original) / sigma_next
            derivative = (self.prev_derivative + derivative) / 2
            dt, sample = self.dt, self.sample
            self.prev_derivative = self.dt = self.sample = None
        return SimpleNamespace(prev_sample=sample + derivative * dt)


class LMSDiscreteSchedulerRef(_SigmaBase):
    """[UPSTREAM] LMSDiscreteScheduler (reference model.py:650-652): linear multistep (order 4) in sigma space, the
    Lagrange-basis integrals by scipy.integrate.quad(epsrel=1e-4) as upstream."""

    def __init__(self, *a, **k):
        k.setdefault("clip_sample", False)
        super().__init__(*a, **k)
        self.set_timesteps(self.config.num_train_timesteps)

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ts, sig = self._interp_sigmas(n)
        self.sigmas = torch.from_numpy(sig)
        self.init_noise_sigma = self.sigmas.max()
        self.timesteps = torch.from_numpy(ts)
        self.derivatives = []

    def scale_model_input(self, sample, timestep):
        i = int((self.timesteps == timestep).nonzero().item())
        return sample / ((self.sigmas[i] ** 2 + 1) ** 0.5)

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

    def step(self, model_output, timestep, sample, order: int = 4, **_):
        i = int((self.timesteps == timestep).nonzero().item())
        sigma = self.sigmas[i]
        pred_original = sample - sigma * model_output
        self.derivatives.append((sample - pred_original) / sigma)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [self.get_lms_coefficient(order, i, c) for c in range(order)]
        prev = sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))
        return SimpleNamespace(prev_sample=prev)


class KarrasVeSchedulerRef:
    """[UPSTREAM] KarrasVeScheduler (reference model.py:685-693, sigma_min 0.01, sigma_max 380, s_churn 80 / 100 / 0):
    stochastic sampler of Karras et al. 2022 Alg. 2.  Upstream's `schedule` holds sigma_max^2 (sigma_min^2/sigma_max^2)^(i/(n-1))
    and the pipeline uses those values as sigma -- reproduced as is."""

    def __init__(self, sigma_min=0.02, sigma_max=100.0, s_noise=1.007, s_churn=80.0, s_min=0.05, s_max=50.0, num_train_timesteps=None):
        self.config = SimpleNamespace(sigma_min=sigma_min, sigma_max=sigma_max, s_noise=s_noise, s_churn=s_churn, s_min=s_min,
                                      s_max=s_max, clip_sample=False)
        self.init_noise_sigma = sigma_max
        self.num_inference_steps = None
        self.timesteps = self.schedule = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ts = np.arange(0, n)[::-1].copy()
        self.timesteps = torch.from_numpy(ts)
        c = self.config
        sched = [c.sigma_max ** 2 * (c.sigma_min ** 2 / c.sigma_max ** 2) ** (i / (n - 1)) for i in self.timesteps]
        self.schedule = torch.tensor(sched, dtype=torch.float32)

    def add_noise_to_input(self, sample, sigma, generator=None, noise=None):
        c = self.config
        gamma = min(c.s_churn / self.num_inference_steps, 2 ** 0.5 - 1) if c.s_min <= sigma <= c.s_max else 0
        eps = c.s_noise * (noise if noise is not None else _randn(sample.shape, generator, sample.device, sample.dtype))
        sigma_hat = sigma + gamma * sigma
        return sample + ((sigma_hat ** 2 - sigma ** 2) ** 0.5 * eps), sigma_hat

    def step(self, model_output, sigma_hat, sigma_prev, sample_hat):
        pred_original = sample_hat + sigma_hat * model_output
        derivative = (sample_hat - pred_original) / sigma_hat
        return SimpleNamespace(prev_sample=sample_hat + (sigma_prev - sigma_hat) * derivative, derivative=derivative,
                               pred_original_sample=pred_original)

    def step_correct(self, model_output, sigma_hat, sigma_prev, sample_hat, sample_prev, derivative):
        pred_original = sample_prev + sigma_prev * model_output
        derivative_corr = (sample_prev - pred_original) / sigma_prev
        return SimpleNamespace(prev_sample=sample_hat + (sigma_prev - sigma_hat) * (0.5 * derivative + 0.5 * derivative_corr),
                               derivative=derivative, pred_original_sample=pred_original)


@torch.no_grad()
def karras_ve_loop(unet, sched: KarrasVeSchedulerRef, init: torch.Tensor, n_steps: int, generator=None, noises=None):
    """[UPSTREAM] KarrasVePipeline.__call__; `init` is the already sigma_max-scaled start (fork contract)."""
    sched.set_timesteps(n_steps)
    x = init
    B = x.shape[0]
    for k, t in enumerate(sched.timesteps):
        sigma = sched.schedule[t]
        sigma_prev = sched.schedule[t - 1] if t > 0 else 0
        x_hat, sigma_hat = sched.add_noise_to_input(x, sigma, generator=generator, noise=None if noises is None else noises[k])
        mo = (sigma_hat / 2) * unet((x_hat + 1) / 2, (sigma_hat / 2).expand(B))[0]
        out = sched.step(mo, sigma_hat, sigma_prev, x_hat)
        if sigma_prev != 0:
            mo = (sigma_prev / 2) * unet((out.prev_sample + 1) / 2, (sigma_prev / 2).expand(B))[0]
            out = sched.step_correct(mo, sigma_hat, sigma_prev, x_hat, out.prev_sample, out.derivative)
        x = out.prev_sample
    return x


class ScoreSdeVeSchedulerRef:
    """S5.  Predictor-corrector VE-SDE sampler."""

    def __init__(self, num_train_timesteps=2000, snr=0.15, sigma_min=0.01, sigma_max=1348.0,
                 sampling_eps=1e-5, correct_steps=1):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, snr=snr, sigma_min=sigma_min,
                                      sigma_max=sigma_max, sampling_eps=sampling_eps, correct_steps=correct_steps,
                                      clip_sample=False)
        self.init_noise_sigma = sigma_max
        self.timesteps = None
        self.set_sigmas(num_train_timesteps)

    def set_timesteps(self, n: int, sampling_eps: float = None):
        eps = sampling_eps if sampling_eps is not None else self.config.sampling_eps
        self.timesteps = torch.linspace(1, eps, n)

    def set_sigmas(self, n: int):
        smin, smax = self.config.sigma_min, self.config.sigma_max
        if self.timesteps is None:
            self.set_timesteps(n)
        self.discrete_sigmas = torch.exp(torch.linspace(math.log(smin), math.log(smax), n))
        self.sigmas = torch.tensor([smin * (smax / smin) ** t for t in self.timesteps])

    def step_correct(self, model_output, sample, generator=None, noise=None):
        z = noise if noise is not None else _randn(sample.shape, generator, sample.device, sample.dtype)
        gnorm = torch.norm(model_output.reshape(model_output.shape[0], -1), dim=-1).mean()
        znorm = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
        step = (self.config.snr * znorm / gnorm) ** 2 * 2
        mean = sample + step * model_output
        return SimpleNamespace(prev_sample=mean + ((step * 2) ** 0.5) * z, prev_sample_mean=mean)

    def step_pred(self, model_output, timestep, sample, generator=None, noise=None):
        tt = timestep * torch.ones(sample.shape[0])
        idx = (tt * (len(self.timesteps) - 1)).long()
        sigma = self.discrete_sigmas[idx]
        adj = torch.where(idx == 0, torch.zeros_like(tt), self.discrete_sigmas[idx - 1])
        diffusion = ((sigma ** 2 - adj ** 2) ** 0.5).flatten()
        while diffusion.dim() < sample.dim():
            diffusion = diffusion.unsqueeze(-1)
        diffusion = diffusion.to(sample.device)
        z = noise if noise is not None else _randn(sample.shape, generator, sample.device, sample.dtype)
        mean = sample + diffusion ** 2 * model_output
        return SimpleNamespace(prev_sample=mean + diffusion * z, prev_sample_mean=mean)


def cosine_with_warmup_lambda(step: int, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """[UPSTREAM] diffusers.optimization.get_cosine_schedule_with_warmup (VillanDiffusion.py:446-450)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


@torch.no_grad()
def sample_loop(unet, sched, init: torch.Tensor, n_steps: int, generator=None, eta: Optional[float] = None,
                start_from: int = 0, noises=None):
    """Fork pipeline contract for the VP samplers: x=init; for t: x=step(unet(x,t),t,x)."""
    sched.set_timesteps(n_steps)
    x = init
    if float(sched.init_noise_sigma) != 1.0:          # sigma-space samplers (Heun / LMSD) start from init * sigma_max
        x = x * sched.init_noise_sigma
    for k, t in enumerate(sched.timesteps[start_from:]):
        if sched.timesteps.dtype.is_floating_point:
            tb = torch.full((x.shape[0],), float(t), dtype=torch.float32)
        else:
            tb = torch.full((x.shape[0],), int(t), dtype=torch.long)
        x_in = sched.scale_model_input(x, t) if hasattr(sched, "scale_model_input") else x
        eps = unet(x_in, tb)[0]
        kw = {}
        if isinstance(sched, DDIMSchedulerRef):
            kw["eta"] = 0.0 if eta is None else eta
        if noises is not None:
            kw["noise"] = noises[k]
        if isinstance(sched, (DDPMSchedulerRef, DDIMSchedulerRef)):
            kw["generator"] = generator
        x = sched.step(eps, t, x, **kw).prev_sample
    return x
