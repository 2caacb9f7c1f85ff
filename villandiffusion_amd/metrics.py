"""Attack-success metrics of the reference's measure() (VillanDiffusion.py:951-1015, 1078-1091): MSE and SSIM of the
backdoor samples against the backdoor target.  SSIM restates torchmetrics' StructuralSimilarityIndexMeasure(data_range=1.0)
defaults: 11x11 gaussian window (sigma 1.5), k1=0.01, k2=0.03, reflect padding with the padded border cropped, mean over the
map per image.  With `device` (what measure() passes: the model's GPU) or CUDA inputs the map is one HIP kernel (`vd_ssim`), so the
whole measure pipeline -- sampling, MSE/SSIM, FID (inception.py), LPIPS (lpips.py) -- runs on the HIP library; CPU tensors without
a device take the torch expression below (host logic, what the CPU tests pin against oracle/metrics_ref.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def mse_batch(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(((a.float() - b.float()) ** 2).flatten(1).mean(1).mean())


def mse_thres_batch(a: torch.Tensor, b: torch.Tensor, thres: float) -> float:
    return float((((a.float() - b.float()) ** 2).flatten(1).mean(1) < thres).float().mean())


def _gauss(k: int, sigma: float) -> torch.Tensor:
    d = torch.arange((1 - k) / 2, (1 + k) / 2, 1.0)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    return (g / g.sum())[None]


def ssim_batch(a: torch.Tensor, b: torch.Tensor, data_range: float = 1.0, k: int = 11, sigma: float = 1.5, device=None,
               chunk: int = 4096) -> float:
    dev = torch.device(device) if device is not None else (a.device if a.device.type == "cuda" else b.device)
    if dev.type == "cuda":
        from . import ops                 # raises when the HIP library is missing: no silent fallback on a GPU box
        g1 = _gauss(k, sigma)
        win = (g1.t() @ g1).contiguous().to(dev)
        c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        tot, n = 0.0, a.shape[0]
        for s0 in range(0, n, chunk):     # `b` is often an expand()ed target: materialise one chunk at a time
            ac = a[s0:s0 + chunk].to(dev).float().contiguous()
            bc = b[s0:s0 + chunk].to(dev).float().contiguous()
            out = torch.empty(ac.shape[0], device=dev, dtype=torch.float32)
            tot += float(ops.ssim(ac, bc, win, out, c1, c2).double().sum())
        return tot / n
    a, b = a.float().cpu(), b.float().cpu()
    C = a.shape[1]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    g1 = _gauss(k, sigma)
    win = (g1.t() @ g1)[None, None].expand(C, 1, k, k)
    p = (k - 1) // 2
    ap, bp = F.pad(a, (p, p, p, p), mode="reflect"), F.pad(b, (p, p, p, p), mode="reflect")
    stack = torch.cat([ap, bp, ap * ap, bp * bp, ap * bp])
    out = F.conv2d(stack, win, groups=C)
    mu_a, mu_b, aa, bb, ab = out.split(a.shape[0])
    va, vb, cab = aa - mu_a ** 2, bb - mu_b ** 2, ab - mu_a * mu_b
    m = ((2 * mu_a * mu_b + c1) * (2 * cab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (va + vb + c2))
    m = m[..., p:-p, p:-p] if m.shape[-1] > 2 * p else m
    return float(m.flatten(1).mean(1).mean())


def activation_statistics(act):
    """(mu, sigma) of a [N, D] activation matrix -- fid_score.py:206-226 (np.mean(axis=0), np.cov(rowvar=False))."""
    import numpy as np
    act = np.asarray(act, dtype=np.float64)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """d^2 = |mu1 - mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2))  -- the FID of two Gaussian activation statistics
    (fid_score.py:150-203, Sutherland's stable version).  The statistics come from InceptionV3 pool3 activations computed on the HIP
    kernels (`inception.py`, `fid_score.py`); `measure()` fills FID when the published pt_inception weight file is present locally
    (`$VILLAN_FID_WEIGHTS`: the reference downloads it, fid_score.py:31-33, and the box has no network) and records None with the reason
    otherwise."""
    import numpy as np
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))
