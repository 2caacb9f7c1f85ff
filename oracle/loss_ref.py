"""CPU oracle: backdoor correction-term tables and the poisoned noise-prediction loss.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference loss.py; PINNED
by tests/golden/loss_tables.npz and tests/golden/loss_batch.npz, which were
produced by importing the reference's loss.py in the build container
(tests/golden/make_golden.py).

Every function cites the reference lines it follows.
"""
from __future__ import annotations

from typing import Tuple

import torch

SDE_VP, SDE_VE, SDE_LDM = "SDE-VP", "SDE-VE", "SDE-LDM"   # model.py:533-535


def hs_vp(alphas: torch.Tensor, alphas_cumprod: torch.Tensor) -> torch.Tensor:
    """loss.py:551-559 -- sequential fp32 recurrence h_i = sqrt(1-abar_i) - sqrt(a_i)(h_{i-1}+res_{i-1})."""
    h = [(1 - alphas_cumprod[0]) ** 0.5]
    res = [torch.zeros(())]
    for i in range(1, len(alphas)):
        res.append((alphas[i] ** 0.5) * (h[i - 1] + res[i - 1]))
        h.append((1 - alphas_cumprod[i]) ** 0.5 - res[i])
    return torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in h])


def ws_ve(sigmas: torch.Tensor) -> torch.Tensor:
    """loss.py:466-474."""
    w = [sigmas[0]]
    res = [torch.zeros(())]
    for i in range(1, len(sigmas)):
        res.append(w[i - 1] ** 2 + res[i - 1])
        w.append((sigmas[i] ** 2 - res[i]) ** 0.5)
    return torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in w])


def hs_ve(rhos_hat: torch.Tensor) -> torch.Tensor:
    """loss.py:476-484."""
    h = [rhos_hat[0]]
    res = [torch.zeros(())]
    for i in range(1, len(rhos_hat)):
        res.append(h[i - 1] + res[i - 1])
        h.append(rhos_hat[i] - res[i])
    return torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in h])


def _by_solver(step, coef, solver_type: str):
    s = str(solver_type).lower()
    if s == "ode":
        return step, 2 * coef
    if s == "sde":
        return step, coef
    raise NotImplementedError(f"Coefficient solver_type: {solver_type} isn't implemented")


def R_coef_vp(alphas_cumprod, alphas, hs=None, psi: float = 1, solver_type: str = "sde",
              vp_scale: float = 1.0, ve_scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """loss.py:561-588 -- (step, coef) tables of length T."""
    bad_step = 1 - alphas_cumprod ** 0.5
    bad_coef = vp_scale * (1 - alphas ** 0.5) * (1 - alphas_cumprod) ** 0.5 / (1 - alphas)
    if psi != 1:
        if hs is None:
            raise ValueError("hs is required when psi != 1")
        troj_step = (1 - alphas_cumprod) ** 0.5
        troj_coef = -ve_scale * ((alphas ** 0.5 - 1) * (1 - alphas_cumprod) ** 0.5 * (1 - alphas)
                                 - hs * (alphas - alphas_cumprod)) / (1 - alphas)
        step = psi * bad_step + (1 - psi) * troj_step
        coef = psi * bad_coef + (1 - psi) * troj_coef
    else:
        step, coef = bad_step, bad_coef
    return _by_solver(step, coef, solver_type)


def R_coef_ve(sigmas, rhos_hat_w: float = 1.0, psi: float = 1, solver_type: str = "sde",
              ve_scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """loss.py:519-549 (the ``_reduce`` form the live code calls, loss.py:902)."""
    if psi != 0:
        raise NotImplementedError("Variance Explode model doesn't support BadDiffusion style correction term")
    prev = torch.roll(sigmas, 1, 0)
    prev[0] = 0
    step = rhos_hat_w * sigmas
    coef = ve_scale * (sigmas * rhos_hat_w / (sigmas + prev))
    return _by_solver(step, coef, solver_type)


class LossFnRef:
    """loss.py:825-1006; loss_type l1 / l2 / huber as loss.py:849-858 (the driver passes "l2", VillanDiffusion.py:1128)."""

    def __init__(self, noise_sched, sde_type: str, loss_type: str = "l2", psi: float = 1, solver_type: str = "sde",
                 vp_scale: float = 1.0, ve_scale: float = 1.0, rhos_hat_w: float = 1.0, rhos_hat_b: float = 0.0):
        self.sched, self.sde_type, self.loss_type = noise_sched, sde_type, loss_type
        self.psi, self.solver_type = psi, solver_type
        self.vp_scale, self.ve_scale, self.rhos_hat_w = vp_scale, ve_scale, rhos_hat_w
        if sde_type in (SDE_VP, SDE_LDM):
            self.alphas, self.alphas_cumprod = noise_sched.alphas, noise_sched.alphas_cumprod
        elif sde_type == SDE_VE:
            self.sigmas = noise_sched.sigmas.flip([0])          # loss.py:834 (ascending)
        else:
            raise NotImplementedError(f"sde_type: {sde_type} isn't implemented")
        self._hs = None
        if loss_type not in ("l1", "l2", "huber"):
            raise NotImplementedError()

    def norm(self, pred, target):
        """loss.py:849-858: elementwise F.l1_loss / F.mse_loss / F.smooth_l1_loss (reduction 'none'); the caller takes .mean()."""
        d = pred - target
        if self.loss_type == "l1":
            return d.abs()
        if self.loss_type == "l2":
            return d ** 2
        a = d.abs()
        return torch.where(a < 1.0, 0.5 * d * d, a - 0.5)

    def tables(self, dtype=torch.float32):
        """loss.py:860-907."""
        if self.sde_type in (SDE_VP, SDE_LDM):
            a, ac = self.alphas.to(dtype), self.alphas_cumprod.to(dtype)
            if self._hs is None:
                self._hs = hs_vp(a, ac)
            return R_coef_vp(ac, a, hs=self._hs.to(dtype), psi=self.psi, solver_type=self.solver_type,
                             vp_scale=self.vp_scale, ve_scale=self.ve_scale)
        return R_coef_ve(self.sigmas.to(dtype), rhos_hat_w=self.rhos_hat_w, psi=self.psi,
                         solver_type=self.solver_type, ve_scale=self.ve_scale)

    def inputs_targets(self, x_start, R, timesteps, noise):
        """loss.py:909-939 -- one formula for clean (R=0) and poisoned samples."""
        n = len(x_start)
        shp = (n,) + (1,) * (x_start.dim() - 1)
        step, coef = self.tables(x_start.dtype)
        coef_t, step_t = coef[timesteps].reshape(shp), step[timesteps].reshape(shp)
        if self.sde_type in (SDE_VP, SDE_LDM):
            x_t = self.sched.add_noise(x_start, noise, timesteps)
        else:
            x_t = x_start + self.sigmas[timesteps].reshape(shp) * noise
        return x_t + step_t * R, coef_t * R + noise

    def p_loss(self, model, x_start, R, timesteps, noise=None):
        """loss.py:978-1006."""
        if len(x_start) == 0:
            return 0
        if noise is None:
            noise = torch.randn_like(x_start)
        x_noisy, target = self.inputs_targets(x_start, R, timesteps, noise)
        if self.sde_type in (SDE_VP, SDE_LDM):
            pred = model(x_noisy.contiguous(), timesteps.contiguous(), return_dict=False)[0]
            return self.norm(pred, target).mean()
        sig = self.sigmas[timesteps]
        pred = model(x_noisy.contiguous(), sig.contiguous(), return_dict=False)[0]
        shp = (len(x_start),) + (1,) * (x_start.dim() - 1)
        return self.norm(-pred * sig.reshape(shp), target).mean()

    def p_loss_by_keys(self, batch, model, target_latent_key, poison_latent_key, timesteps, noise=None, **_):
        """loss.py:972-976 (vae=None path: latents are precomputed, VillanDiffusion.py:1159)."""
        return self.p_loss(model, batch[target_latent_key], batch[poison_latent_key], timesteps, noise)
