"""FID feature extractor on the HIP kernels: the `InceptionV3` wrapper of pytorch-fid 0.3.0 (requirement.txt:151) that the reference's
fid_score.py:50,264-270 builds as `InceptionV3([block_idx])` -- forward only, no-grad, inference BatchNorm.

What runs where:
  * every BasicConv2d (conv, bias-free -> BatchNorm(eps 1e-3, running statistics) -> ReLU) is ONE vd_gemm launch: BatchNorm is folded into
    the weights and a bias once at load time (w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps)), ReLU is the
    epilogue's `act`; 1x1 convolutions are plain GEMMs, the 3x3 / 5x5 / 1x7 / 7x1 / stride-2 ones the general gather mode VD_B_CONVG
    (exact-f32 MFMA: the FID statistics are taken at full precision);
  * the branch outputs of a Mixed block are written straight into their channel slice of the block's output buffer (no torch.cat);
  * 3x3 max / average pooling (padding excluded from the divisor, as pytorch-fid's FID variants) = vd_pool3; the 299 x 299 bilinear resize
    with the 2x - 1 input scaling = vd_resize_bilinear; the final global average = vd_rowsum.
State-dict keys are torchvision's Inception3 names, so the published `pt_inception-2015-12-05-6726825d.pth` loads unchanged (`fc.*` ignored).
"""
from __future__ import annotations

import os
from typing import Dict, Sequence

import torch

from . import ops

FID_WEIGHTS_FILE = "pt_inception-2015-12-05-6726825d.pth"        # pytorch-fid's FID_WEIGHTS_URL basename


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _Conv:
    """BasicConv2d with BatchNorm folded in."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, dev, stride=1, padding=0, eps=1e-3):
        w = sd[prefix + ".conv.weight"].float()
        g, b = sd[prefix + ".bn.weight"].float(), sd[prefix + ".bn.bias"].float()
        mean, var = sd[prefix + ".bn.running_mean"].float(), sd[prefix + ".bn.running_var"].float()
        scale = g / torch.sqrt(var + eps)
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.w2d = (w * scale[:, None, None, None]).reshape(self.cout, -1).contiguous().to(dev)
        self.bias = (b - mean * scale).contiguous().to(dev)
        self.stride = stride
        self.ph, self.pw = _pair(padding)

    def out_hw(self, H, W):
        return (H + 2 * self.ph - self.kh) // self.stride + 1, (W + 2 * self.pw - self.kw) // self.stride + 1

    def __call__(self, x, out=None):
        B, _, H, W = x.shape
        if out is None:
            out = torch.empty((B, self.cout) + self.out_hw(H, W), device=x.device, dtype=torch.float32)
        ops.conv2d_general(x, self.w2d, self.bias, out, self.kh, self.kw, self.stride, self.ph, self.pw, relu=True)
        return out


def _pool(x, stride, pad, mode, out=None):
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty((B, C, (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1), device=x.device, dtype=torch.float32)
    ops.pool3(x, out, stride=stride, pad=pad, mode=mode)
    return out


class InceptionV3:
    """`model(batch)[0]` is the pool3 activation [B, 2048, 1, 1] for the default block (fid_score.py:132)."""
    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks: Sequence[int] = (3,), resize_input: bool = True, normalize_input: bool = True, state_dict=None,
                 device=None):
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, "Last possible output block index is 3"
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if state_dict is None:
            state_dict = load_fid_weights()
        self.load_state_dict(state_dict)

    # ---- weights ----
    def load_state_dict(self, sd):
        dev = self.device
        c = lambda name, **kw: _Conv(sd, name, dev, **kw)                                       # noqa: E731
        self.stem = [c("Conv2d_1a_3x3", stride=2), c("Conv2d_2a_3x3"), c("Conv2d_2b_3x3", padding=1), c("Conv2d_3b_1x1"), c("Conv2d_4a_3x3")]
        A = lambda p: dict(b1=c(p + ".branch1x1"), b5=[c(p + ".branch5x5_1"), c(p + ".branch5x5_2", padding=2)],           # noqa: E731
                           bd=[c(p + ".branch3x3dbl_1"), c(p + ".branch3x3dbl_2", padding=1), c(p + ".branch3x3dbl_3", padding=1)],
                           bp=c(p + ".branch_pool"))
        self.m5 = [A("Mixed_5b"), A("Mixed_5c"), A("Mixed_5d")]
        self.m6a = dict(b3=c("Mixed_6a.branch3x3", stride=2),
                        bd=[c("Mixed_6a.branch3x3dbl_1"), c("Mixed_6a.branch3x3dbl_2", padding=1), c("Mixed_6a.branch3x3dbl_3", stride=2)])
        Cb = lambda p: dict(b1=c(p + ".branch1x1"),                                                                          # noqa: E731
                            b7=[c(p + ".branch7x7_1"), c(p + ".branch7x7_2", padding=(0, 3)), c(p + ".branch7x7_3", padding=(3, 0))],
                            bd=[c(p + ".branch7x7dbl_1"), c(p + ".branch7x7dbl_2", padding=(3, 0)), c(p + ".branch7x7dbl_3", padding=(0, 3)),
                                c(p + ".branch7x7dbl_4", padding=(3, 0)), c(p + ".branch7x7dbl_5", padding=(0, 3))],
                            bp=c(p + ".branch_pool"))
        self.m6 = [Cb("Mixed_6b"), Cb("Mixed_6c"), Cb("Mixed_6d"), Cb("Mixed_6e")]
        self.m7a = dict(b3=[c("Mixed_7a.branch3x3_1"), c("Mixed_7a.branch3x3_2", stride=2)],
                        b7=[c("Mixed_7a.branch7x7x3_1"), c("Mixed_7a.branch7x7x3_2", padding=(0, 3)), c("Mixed_7a.branch7x7x3_3", padding=(3, 0)),
                            c("Mixed_7a.branch7x7x3_4", stride=2)])
        E = lambda p, pool: dict(b1=c(p + ".branch1x1"), b3_1=c(p + ".branch3x3_1"), b3_2a=c(p + ".branch3x3_2a", padding=(0, 1)),      # noqa: E731
                                 b3_2b=c(p + ".branch3x3_2b", padding=(1, 0)), bd_1=c(p + ".branch3x3dbl_1"),
                                 bd_2=c(p + ".branch3x3dbl_2", padding=1), bd_3a=c(p + ".branch3x3dbl_3a", padding=(0, 1)),
                                 bd_3b=c(p + ".branch3x3dbl_3b", padding=(1, 0)), bp=c(p + ".branch_pool"), pool=pool)
        self.m7 = [E("Mixed_7b", "avg"), E("Mixed_7c", "max")]       # FIDInceptionE_1 / FIDInceptionE_2
        return self

    def eval(self):
        return self

    def to(self, device=None):
        return self

    # ---- blocks: every branch writes its channel slice of `out` ----
    @staticmethod
    def _chain(convs, x, out):
        for cv in convs[:-1]:
            x = cv(x)
        return convs[-1](x, out)

    def _A(self, m, x):
        B, _, H, W = x.shape
        pf = m["bp"].cout
        out = torch.empty((B, 64 + 64 + 96 + pf, H, W), device=x.device, dtype=torch.float32)
        m["b1"](x, out[:, 0:64])
        self._chain(m["b5"], x, out[:, 64:128])
        self._chain(m["bd"], x, out[:, 128:224])
        m["bp"](_pool(x, 1, 1, "avg"), out[:, 224:224 + pf])
        return out

    def _B(self, m, x):
        B, C, H, W = x.shape
        OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        out = torch.empty((B, 384 + 96 + C, OH, OW), device=x.device, dtype=torch.float32)
        m["b3"](x, out[:, 0:384])
        self._chain(m["bd"], x, out[:, 384:480])
        _pool(x, 2, 0, "max", out[:, 480:480 + C])
        return out

    def _C(self, m, x):
        B, _, H, W = x.shape
        out = torch.empty((B, 768, H, W), device=x.device, dtype=torch.float32)
        m["b1"](x, out[:, 0:192])
        self._chain(m["b7"], x, out[:, 192:384])
        self._chain(m["bd"], x, out[:, 384:576])
        m["bp"](_pool(x, 1, 1, "avg"), out[:, 576:768])
        return out

    def _D(self, m, x):
        B, C, H, W = x.shape
        OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        out = torch.empty((B, 320 + 192 + C, OH, OW), device=x.device, dtype=torch.float32)
        self._chain(m["b3"], x, out[:, 0:320])
        self._chain(m["b7"], x, out[:, 320:512])
        _pool(x, 2, 0, "max", out[:, 512:512 + C])
        return out

    def _E(self, m, x):
        B, _, H, W = x.shape
        out = torch.empty((B, 2048, H, W), device=x.device, dtype=torch.float32)
        m["b1"](x, out[:, 0:320])
        b3 = m["b3_1"](x)
        m["b3_2a"](b3, out[:, 320:704])
        m["b3_2b"](b3, out[:, 704:1088])
        bd = m["bd_2"](m["bd_1"](x))
        m["bd_3a"](bd, out[:, 1088:1472])
        m["bd_3b"](bd, out[:, 1472:1856])
        m["bp"](_pool(x, 1, 1, m["pool"]), out[:, 1856:2048])
        return out

    @torch.no_grad()
    def __call__(self, inp: torch.Tensor):
        x = inp.to(self.device).float().contiguous()
        B = x.shape[0]
        mul, add = (2.0, -1.0) if self.normalize_input else (1.0, 0.0)
        if self.resize_input:
            x = ops.resize_bilinear(x, torch.empty((B, x.shape[1], 299, 299), device=self.device, dtype=torch.float32), mul, add)
        elif self.normalize_input:
            x = ops.resize_bilinear(x, torch.empty_like(x), mul, add)          # same size: the interpolation is the identity
        outs = []

        def emit(idx, t):
            if idx in self.output_blocks:
                outs.append(t)
            return idx == self.last_needed_block

        s = self.stem
        x = _pool(s[2](s[1](s[0](x))), 2, 0, "max")
        if emit(0, x):
            return outs
        x = _pool(s[4](s[3](x)), 2, 0, "max")
        if emit(1, x):
            return outs
        for m in self.m5:
            x = self._A(m, x)
        x = self._B(self.m6a, x)
        for m in self.m6:
            x = self._C(m, x)
        if emit(2, x):
            return outs
        x = self._D(self.m7a, x)
        for m in self.m7:
            x = self._E(m, x)
        Bc, C, H, W = x.shape
        pooled = torch.empty((Bc, C), device=self.device, dtype=torch.float32)
        ops.rowsum(x, pooled)                                        # sum over the H*W pixels of every (image, channel)
        ops.scale_(pooled, 1.0 / (H * W))
        emit(3, pooled.view(Bc, C, 1, 1))
        return outs


def load_fid_weights(path: str = None):
    """The FID InceptionV3 weights from a LOCAL file (there is no network on the box): `path`, $VILLAN_FID_WEIGHTS, or
    $VILLAN_CKPT_ROOT/pt_inception-2015-12-05-6726825d.pth (what pytorch-fid downloads from its FID_WEIGHTS_URL)."""
    cands = [path, os.environ.get("VILLAN_FID_WEIGHTS"),
             os.path.join(os.environ.get("VILLAN_CKPT_ROOT", os.path.expanduser("~/.cache/torch/hub/checkpoints")), FID_WEIGHTS_FILE)]
    for c in cands:
        if c and os.path.isfile(c):
            return torch.load(c, map_location="cpu")
    raise FileNotFoundError(
        f"FID InceptionV3 weights not found (looked at {[c for c in cands if c]}): download {FID_WEIGHTS_FILE} "
        f"(pytorch-fid's FID_WEIGHTS_URL) and point $VILLAN_FID_WEIGHTS at it")
