"""CPU oracle (TEST INFRASTRUCTURE ONLY) of `lpips.LPIPS(net='alex')` (lpips==0.1.4, requirement.txt:84), which the reference's
measure_inpaint() calls as `float(torch.mean(lpips.LPIPS(net='alex').to(device)(recover_imgs, target_imgs)))` (VillanDiffusion.py:892).

**Parity unpinned**: lpips (and torchvision's AlexNet it wraps) is a third-party dependency that is not under /root/reference and not
installed here; this file restates its published computation (version '0.1', lpips=True, spatial=False) in plain fp32 torch:

* ScalingLayer: (x - shift) / scale with shift = (-.030, -.088, -.188), scale = (.458, .448, .450); inputs are taken as they come
  (`normalize=False`: the reference hands it [0, 1] images although the metric expects [-1, 1] -- reproduced as called);
* torchvision AlexNet `features`: Conv(3, 64, k11, s4, p2) ReLU | MaxPool(3, 2) Conv(64, 192, k5, p2) ReLU | MaxPool(3, 2)
  Conv(192, 384, k3, p1) ReLU | Conv(384, 256, k3, p1) ReLU | Conv(256, 256, k3, p1) ReLU -- the five ReLU outputs are the taps;
* per tap: unit-normalise over channels (x / (sqrt(sum_c x^2) + 1e-10)), squared difference, 1x1 convolution to one channel without
  bias (`lin{k}.model.1.weight` [1, C, 1, 1]; its Dropout is the identity in eval), spatial mean; the score is the sum over the taps,
  shape [N, 1, 1, 1].
Known answers that pin the restatement: 2 469 696 parameters in the five convolutions (torchvision AlexNet has 61 100 840 with its
classifier: 58 631 144), tap widths (64, 192, 384, 256, 256), d(x, x) = 0, symmetry, non-negativity for non-negative lin weights.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)
CHNS = (64, 192, 384, 256, 256)


class AlexFeaturesRef(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(), nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(), nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(),
            nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(),
            nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU())        # the trailing MaxPool of torchvision's `features` is never reached

    def forward(self, x):
        taps = []
        for i, m in enumerate(self.features):
            x = m(x)
            if i in (1, 4, 7, 9, 11):
                taps.append(x)
        return taps


class LPIPSRef(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = AlexFeaturesRef()
        self.lins = nn.ModuleList([nn.Conv2d(c, 1, 1, bias=False) for c in CHNS])
        self.register_buffer("shift", torch.tensor(SHIFT)[None, :, None, None])
        self.register_buffer("scale", torch.tensor(SCALE)[None, :, None, None])
        self.eval()

    def randomize(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.net.features:
                if isinstance(m, nn.Conv2d):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / m.weight[0].numel()) ** 0.5)
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            for l in self.lins:
                l.weight.copy_(torch.rand(l.weight.shape, generator=g) * 0.1)      # the published lin weights are non-negative
        return self

    def flat_state_dict(self):
        """Keys as the two published files use them: torchvision `features.N.weight|bias` + lpips `linK.model.1.weight`."""
        sd = {f"features.{k}": v for k, v in self.net.features.state_dict().items()}
        sd.update({f"lin{k}.model.1.weight": l.weight.detach() for k, l in enumerate(self.lins)})
        return sd

    @torch.no_grad()
    def forward(self, in0, in1, normalize=False):
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        f0, f1 = self.net((in0 - self.shift) / self.scale), self.net((in1 - self.shift) / self.scale)
        val = 0
        for k in range(5):
            n0 = f0[k] / (torch.sqrt(torch.sum(f0[k] ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = f1[k] / (torch.sqrt(torch.sum(f1[k] ** 2, dim=1, keepdim=True)) + 1e-10)
            val = val + self.lins[k]((n0 - n1) ** 2).mean([2, 3], keepdim=True)
        return val
