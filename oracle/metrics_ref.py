"""CPU oracle of the attack-success scores of the reference's measure() (VillanDiffusion.py:951-1015, 1078-1091).

TEST INFRASTRUCTURE (see oracle/__init__.py).  MSE is `nn.MSELoss(reduction='none')(...).mean(dim=[1,2,3])` averaged over the batch
(VillanDiffusion.py:963-966).  SSIM is torchmetrics' StructuralSimilarityIndexMeasure(data_range=1.0) (VillanDiffusion.py:1001-1007):
torchmetrics is not installed here, so PARITY UNPINNED -- this file restates the published algorithm (Wang et al. 2004 as torchmetrics
implements it: 11 x 11 gaussian window, sigma 1.5, k1 0.01, k2 0.03, reflect padding whose border is cropped from the map, mean over
the map, mean over the batch) with plain numpy loops, independently of villandiffusion_amd/metrics.py (which uses one grouped
convolution)."""
import numpy as np


def mse_ref(a: np.ndarray, b: np.ndarray) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(((a - b) ** 2).reshape(a.shape[0], -1).mean(1).mean())


def _gauss1d(k=11, sigma=1.5):
    d = np.arange((1 - k) / 2, (1 + k) / 2, 1.0)
    g = np.exp(-((d / sigma) ** 2) / 2)
    return g / g.sum()


def _filter(img2d, g):
    """'valid' separable gaussian filter of one padded channel."""
    k = len(g)
    H, W = img2d.shape
    tmp = np.zeros((H, W - k + 1))
    for j in range(k):
        tmp += g[j] * img2d[:, j:j + W - k + 1]
    out = np.zeros((H - k + 1, W - k + 1))
    for i in range(k):
        out += g[i] * tmp[i:i + H - k + 1, :]
    return out


def ssim_ref(a: np.ndarray, b: np.ndarray, data_range: float = 1.0) -> float:
    """a, b: [N, C, H, W] in [0, data_range]."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    g = _gauss1d()
    p = (len(g) - 1) // 2
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    per_image = []
    for n in range(a.shape[0]):
        maps = []
        for c in range(a.shape[1]):
            x, y = np.pad(a[n, c], p, mode="reflect"), np.pad(b[n, c], p, mode="reflect")
            mx, my = _filter(x, g), _filter(y, g)
            vx, vy, cxy = _filter(x * x, g) - mx * mx, _filter(y * y, g) - my * my, _filter(x * y, g) - mx * my
            m = ((2 * mx * my + c1) * (2 * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))
            maps.append(m[p:-p, p:-p] if m.shape[-1] > 2 * p else m)          # the reflect-padded border is cropped from the map
        per_image.append(np.mean(maps))
    return float(np.mean(per_image))
