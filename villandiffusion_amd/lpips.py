"""`lpips.LPIPS(net='alex')` on the HIP kernels -- the perceptual score of the denoise / inpaint measure tasks (reference
VillanDiffusion.py:337,892: `float(torch.mean(lpips.LPIPS(net='alex').to(device)(recover_imgs, target_imgs)))`).

Forward only.  The five AlexNet convolutions (+ bias + ReLU) are vd_gemm launches in the general gather mode (11x11 stride 4, 5x5, 3x3),
the two max-poolings vd_pool3, the input ScalingLayer vd_channel_affine, and each tap's unit-normalise / squared difference / 1x1 "lin"
convolution / spatial mean is ONE vd_lpips_layer launch per tap.  Inputs are taken as the caller hands them (`normalize=False`, as the
reference calls it).  Weights come from two LOCAL files (no network on the box): torchvision's `alexnet-owt-7be5be79.pth`
(`features.N.weight|bias`) and the lpips package's `weights/v0.1/alex.pth` (`linK.model.1.weight`).
"""
from __future__ import annotations

import os

import torch

from . import ops

ALEXNET_FILE, LIN_FILE = "alexnet-owt-7be5be79.pth", "alex.pth"
SHIFT, SCALE = (-.030, -.088, -.188), (.458, .448, .450)
_CONVS = ((0, 11, 4, 2), (3, 5, 1, 2), (6, 3, 1, 1), (8, 3, 1, 1), (10, 3, 1, 1))          # features index, kernel, stride, padding


def load_lpips_weights(alexnet_path: str = None, lin_path: str = None):
    root = os.environ.get("VILLAN_CKPT_ROOT", os.path.expanduser("~/.cache/torch/hub/checkpoints"))
    a = alexnet_path or os.environ.get("VILLAN_ALEXNET_WEIGHTS") or os.path.join(root, ALEXNET_FILE)
    l = lin_path or os.environ.get("VILLAN_LPIPS_WEIGHTS") or os.path.join(root, "lpips_" + LIN_FILE)
    for p in (a, l):
        if not os.path.isfile(p):
            raise FileNotFoundError(f"LPIPS weights not found at {p}: needs torchvision's {ALEXNET_FILE} ($VILLAN_ALEXNET_WEIGHTS) and the lpips "
                                    f"package's weights/v0.1/{LIN_FILE} ($VILLAN_LPIPS_WEIGHTS) as local files")
    sd = dict(torch.load(a, map_location="cpu"))
    sd.update(torch.load(l, map_location="cpu"))
    return sd


class LPIPS:
    def __init__(self, net: str = "alex", state_dict=None, device=None):
        if net != "alex":
            raise NotImplementedError("only net='alex' (what the reference uses) is implemented")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        sd = state_dict if state_dict is not None else load_lpips_weights()
        dev = self.device
        self.convs = []
        for idx, k, s, p in _CONVS:
            w = sd[f"features.{idx}.weight"].float()
            self.convs.append((w.reshape(w.shape[0], -1).contiguous().to(dev), sd[f"features.{idx}.bias"].float().contiguous().to(dev), k, s, p))
        self.lins = [sd[f"lin{k}.model.1.weight"].float().reshape(-1).contiguous().to(dev) for k in range(5)]
        self.mul = torch.tensor([1.0 / s for s in SCALE], device=dev)
        self.add = torch.tensor([-sh / sc for sh, sc in zip(SHIFT, SCALE)], device=dev)

    def to(self, device=None):
        return self

    def eval(self):
        return self

    def _conv(self, x, i):
        w2d, b, k, s, p = self.convs[i]
        B, _, H, W = x.shape
        out = torch.empty((B, w2d.shape[0], (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1), device=x.device, dtype=torch.float32)
        return ops.conv2d_general(x, w2d, b, out, k, k, s, p, p, relu=True)

    @staticmethod
    def _pool(x):
        B, C, H, W = x.shape
        out = torch.empty((B, C, (H - 3) // 2 + 1, (W - 3) // 2 + 1), device=x.device, dtype=torch.float32)
        ops.pool3(x, out, stride=2, pad=0, mode="max")
        return out

    def features(self, x):
        if x.shape[1] == 1:               # the reference ScalingLayer broadcasts a 1-channel input (MNIST tasks) against its [1,3,1,1] shift / scale
            x = x.expand(-1, 3, -1, -1).contiguous()
        x = ops.channel_affine(x, self.mul, self.add, torch.empty_like(x))
        t1 = self._conv(x, 0)
        t2 = self._conv(self._pool(t1), 1)
        t3 = self._conv(self._pool(t2), 2)
        t4 = self._conv(t3, 3)
        t5 = self._conv(t4, 4)
        return [t1, t2, t3, t4, t5]

    @torch.no_grad()
    def __call__(self, in0: torch.Tensor, in1: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        a, b = in0.to(self.device).float().contiguous(), in1.to(self.device).float().contiguous()
        if normalize:
            a, b = 2 * a - 1, 2 * b - 1
        f0, f1 = self.features(a), self.features(b)
        out = torch.empty(a.shape[0], device=self.device, dtype=torch.float32)
        for k in range(5):
            ops.lpips_layer(f0[k], f1[k], self.lins[k], out, accumulate=k > 0)
        return out.view(-1, 1, 1, 1)

    forward = __call__
