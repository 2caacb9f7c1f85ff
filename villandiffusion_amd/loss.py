"""Backdoor (poisoned) noise-prediction loss on MI355X -- same surface as the reference's ``loss.LossFn``
(loss.py:825-1006) and the coefficient helpers (loss.py:466-588).

What changed under the surface
* the (step, coef) correction tables (L1-L3 in SURVEY.md §8a) are built ONCE on the host with the reference's fp32 op
  sequence and kept on the device (the reference rebuilds them every step, loss.py:917);
* q-sample + backdoor shift + target (L4) is ONE kernel (``vd_qsample_backdoor``) and MSE forward+backward (L5)
  is ONE kernel pair (``vd_mse_fwd_bwd``); ``loss.backward()`` then drives the UNet's explicit backward.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import ops

SDE_VP, SDE_VE, SDE_LDM = "SDE-VP", "SDE-VE", "SDE-LDM"


# ---------------------------------------------------------------------------------------------- coefficient tables
def get_hs_vp(alphas: torch.Tensor, alphas_cumprod: torch.Tensor) -> torch.Tensor:
    """h_0 = sqrt(1-abar_0); res_i = sqrt(a_i)(h_{i-1}+res_{i-1}); h_i = sqrt(1-abar_i) - res_i  (loss.py:551-559).
    Sequential recurrence on fp32 0-d CPU tensors -- deliberately the SAME torch scalar kernels the reference runs
    (torch's CPU `x ** 0.5` is not always the correctly rounded sqrt, so e.g. numpy float32 would differ in the last bit).
    Runs once per LossFn (the reference caches it too, loss.py:872-874)."""
    a, ac = alphas.detach().float().cpu(), alphas_cumprod.detach().float().cpu()
    h = [(1 - ac[0]) ** 0.5]
    res = torch.zeros(())
    for i in range(1, len(a)):
        res = (a[i] ** 0.5) * (h[i - 1] + res)
        h.append((1 - ac[i]) ** 0.5 - res)
    return torch.stack(h)


def get_ws_ve(sigmas: torch.Tensor) -> torch.Tensor:
    """loss.py:466-474."""
    s_ = sigmas.detach().float().cpu()
    w = [s_[0]]
    res = torch.zeros(())
    for i in range(1, len(s_)):
        res = w[i - 1] ** 2 + res
        w.append((s_[i] ** 2 - res) ** 0.5)
    return torch.stack(w)


def get_hs_ve(rhos_hat: torch.Tensor) -> torch.Tensor:
    """loss.py:476-484."""
    r = rhos_hat.detach().float().cpu()
    h = [r[0]]
    res = torch.zeros(())
    for i in range(1, len(r)):
        res = h[i - 1] + res
        h.append(r[i] - res)
    return torch.stack(h)


def _solver(step, coef, solver_type):
    s = str(solver_type).lower()
    if s == "ode":
        return step, 2 * coef
    if s == "sde":
        return step, coef
    raise NotImplementedError(f"Coefficient solver_type: {solver_type} isn't implemented")


def get_R_coef_gen_vp(alphas_cumprod, alphas, hs=None, psi: float = 1, solver_type: str = "sde", vp_scale: float = 1.0,
                      ve_scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """BadDiffusion (psi=1) / TrojDiff (psi=0) correction tables and their psi-blend (loss.py:561-588)."""
    one_m_ac_sqrt = (1 - alphas_cumprod) ** 0.5
    step_b = 1 - alphas_cumprod ** 0.5
    coef_b = vp_scale * (1 - alphas ** 0.5) * one_m_ac_sqrt / (1 - alphas)
    if psi == 1:
        return _solver(step_b, coef_b, solver_type)
    if hs is None:
        raise ValueError(f"Argument hs shouldn't be {hs} when psi is {psi}")
    coef_t = -ve_scale * ((alphas ** 0.5 - 1) * one_m_ac_sqrt * (1 - alphas) - hs * (alphas - alphas_cumprod)) / (1 - alphas)
    return _solver(psi * step_b + (1 - psi) * one_m_ac_sqrt, psi * coef_b + (1 - psi) * coef_t, solver_type)


def get_R_coef_gen_ve_reduce(sigmas, hs=None, rhos_hat_w: float = 1.0, psi: float = 1, solver_type: str = "sde",
                             vp_scale: float = 1.0, ve_scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """loss.py:519-549."""
    if psi != 0:
        raise NotImplementedError("Variance Explode model doesn't support BadDiffusion style correction term")
    prev = torch.roll(sigmas, 1, 0)
    prev[0] = 0
    return _solver(rhos_hat_w * sigmas, ve_scale * (sigmas * rhos_hat_w / (sigmas + prev)), solver_type)


# ---------------------------------------------------------------------------------------------- fused MSE
class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, y, pscale, gscale, partial, kind="l2"):
        dpred = torch.empty_like(pred)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        ops.mse_fwd_bwd(pred.contiguous(), y, dpred, loss, partial, pscale=pscale, gscale=gscale, kind=kind)
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None, None, None


class LossFn:
    """Drop-in for the reference's ``LossFn`` (same constructor and ``p_loss`` / ``p_loss_by_keys`` signatures)."""
    RANDN_BOUND: float = 2.5

    def __init__(self, noise_sched, sde_type: str, loss_type: str = "l2", psi: float = 1, solver_type: str = "sde",
                 vp_scale: float = 1.0, ve_scale: float = 1.0, rhos_hat_w: float = 1.0, rhos_hat_b: float = 0.0):
        if sde_type not in (SDE_VP, SDE_VE, SDE_LDM):
            raise NotImplementedError(f"sde_type: {sde_type} isn't implemented")
        if loss_type not in ops.LOSS_KINDS:       # loss.py:849-858: l1 / l2 / huber, anything else raises
            raise NotImplementedError(f"loss_type: {loss_type} isn't implemented")
        self._loss_type = loss_type
        self._sched, self._sde, self._psi, self._solver = noise_sched, sde_type, psi, solver_type
        # the schedule is captured HERE like the reference does (loss.py:829-834): a VE pipeline's set_sigmas(n) later
        # replaces noise_sched.sigmas by the n-step inference table, which must not shrink the training tables
        if sde_type == SDE_VE:
            self._sigmas_asc = noise_sched.sigmas.flip([0]).float().cpu().clone()
        else:
            self._alphas = noise_sched.alphas.float().cpu().clone()
            self._alphas_cumprod = noise_sched.alphas_cumprod.float().cpu().clone()
        self._vp_scale, self._ve_scale, self._w, self._b = vp_scale, ve_scale, rhos_hat_w, rhos_hat_b
        self._dev_tabs = None
        self._partial = None
        self.grad_scale = 1.0            # trainer sets 1/gradient_accumulation_steps (accelerator.backward semantics)
        self.noise_seed: Optional[int] = None
        self._noise_off = 0

    # host tables (fp32, CPU) -- loss.py:860-907
    def get_R_step_coef(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._sde in (SDE_VP, SDE_LDM):
            a, ac = self._alphas, self._alphas_cumprod
            hs = get_hs_vp(a, ac) if self._psi != 1 else None
            return get_R_coef_gen_vp(ac, a, hs=hs, psi=self._psi, solver_type=self._solver, vp_scale=self._vp_scale,
                                     ve_scale=self._ve_scale)
        sig = self._sigmas_asc
        return get_R_coef_gen_ve_reduce(sig, hs=True, rhos_hat_w=self._w, psi=self._psi, solver_type=self._solver,
                                        ve_scale=self._ve_scale)

    def _tables(self, dev):
        if self._dev_tabs is None or self._dev_tabs[0].device != dev:
            step, coef = self.get_R_step_coef()
            if self._sde in (SDE_VP, SDE_LDM):
                ac = self._alphas_cumprod
                ta, ts = ac ** 0.5, (1 - ac) ** 0.5
                self._dev_tabs = (step.to(dev), coef.to(dev), ta.to(dev), ts.to(dev))
            else:
                self._dev_tabs = (step.to(dev), coef.to(dev), None, self._sigmas_asc.to(dev))
            self._partial = torch.empty(1024, device=dev, dtype=torch.float32)
        return self._dev_tabs

    def get_inputs_targets(self, x_start, R, timesteps, noise):
        """(x_t, y) of loss.py:909-939 in one kernel."""
        dev = x_start.device
        step, coef, ta, ts = self._tables(dev)
        x_t, y = torch.empty_like(x_start), torch.empty_like(x_start)
        ops.qsample_backdoor(x_start.contiguous(), R.contiguous(), noise.contiguous(),
                             timesteps.to(device=dev, dtype=torch.int64).contiguous(), ta, ts, step, coef, x_t, y)
        return x_t, y

    def _noise(self, x):
        if self.noise_seed is None:
            self.noise_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        z = torch.empty_like(x)
        ops.randn(z, self.noise_seed, self._noise_off)
        self._noise_off += (x.numel() + 3) // 4
        return z

    def p_loss(self, model, x_start: torch.Tensor, R: torch.Tensor, timesteps: torch.Tensor, noise: torch.Tensor = None):
        if len(x_start) == 0:
            return 0
        dev = model.device if hasattr(model, "device") else x_start.device
        x_start, R = x_start.to(dev).float(), R.to(dev).float()
        timesteps = timesteps.to(dev)
        if noise is None:
            noise = self._noise(x_start)
        x_t, y = self.get_inputs_targets(x_start, R, timesteps, noise.to(dev))
        if self._sde in (SDE_VP, SDE_LDM):
            pred = model(x_t, timesteps.contiguous(), return_dict=False)[0]
            return _MSE.apply(pred, y, None, self.grad_scale, self._partial, self._loss_type)
        sig_t = self._dev_tabs[3][timesteps]
        pred = model(x_t, sig_t.contiguous(), return_dict=False)[0]
        return _MSE.apply(pred, y, (-sig_t).contiguous(), self.grad_scale, self._partial, self._loss_type)

    def p_loss_by_keys(self, batch, model, target_latent_key, poison_latent_key, timesteps, vae=None, noise=None,
                       weight_dtype=None, scaling_factor=None):
        x0, R = batch[target_latent_key], batch[poison_latent_key]
        if vae is not None:          # loss.py:942-970: encode on the fly (training itself passes vae=None + precomputed latents)
            x0, R = self.encode_latents(vae, x0, scaling_factor), self.encode_latents(vae, R, scaling_factor)
        return self.p_loss(model=model, x_start=x0, R=R, timesteps=timesteps, noise=noise)

    @staticmethod
    def encode_latents(vae, x: torch.Tensor, scaling_factor: Optional[float] = None) -> torch.Tensor:
        """loss.py:941-950  vae.encode(x).latents (* scaling_factor), detached."""
        lat = vae.encode(x).latents
        if scaling_factor is not None:
            lat = lat * scaling_factor
        return lat.detach()

    @staticmethod
    def decode_latents(vae, x: torch.Tensor, scaling_factor: Optional[float] = None) -> torch.Tensor:
        """loss.py:951-962  vae.decode(x).sample (/ scaling_factor), detached."""
        out = vae.decode(x).sample
        if scaling_factor is not None:
            out = out / scaling_factor
        return out.detach()
