"""CPU oracle: the NCSN++ score network (diffusers ``UNet2DModel`` with ``Skip*Block2D`` blocks, Fourier time embedding and
FIR resampling) the reference uses for SDE-VE (model.py:839-857, 876-894; checkpoints ``fusing/cifar10-ncsnpp-ve``,
``google/ncsnpp-celebahq-256``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in the un-vendored diffusers fork
(requirement.txt:37); this restates the published upstream modules (diffusers ~0.16: ``GaussianFourierProjection``,
``ResnetBlock2D`` with ``up/down`` FIR kernels and ``output_scale_factor``, ``AttentionBlock`` with
``rescale_output_factor``, ``SkipDownBlock2D / AttnSkipDownBlock2D / SkipUpBlock2D / AttnSkipUpBlock2D``,
``upfirdn2d_native``) with diffusers state-dict names.  Checked by known-answer tests in tests/test_ncsnpp.py (FIR
resampling identities, shapes, state-dict surface, sigma scaling of the output).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import TimestepEmbedding, _ToOut

SQRT2 = float(np.sqrt(2.0))


# ------------------------------------------------------------------------------------------------ FIR resampling
def upfirdn2d_native(x, kernel, up=1, down=1, pad=(0, 0)):
    """[UPSTREAM] upfirdn2d_native: zero-stuff by `up`, pad, correlate with the FLIPPED kernel, decimate by `down`."""
    p0, p1 = pad
    b, c, h, w = x.shape
    t = x.reshape(-1, h, w, 1)
    kh, kw = kernel.shape
    out = t.view(-1, h, 1, w, 1, 1)
    out = F.pad(out, [0, 0, 0, up - 1, 0, 0, 0, up - 1])
    out = out.view(-1, h * up, w * up, 1)
    out = F.pad(out, [0, 0, max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    out = out[:, max(-p0, 0): out.shape[1] - max(-p1, 0), max(-p0, 0): out.shape[2] - max(-p1, 0), :]
    out = out.permute(0, 3, 1, 2)
    out = out.reshape([-1, 1, h * up + p0 + p1, w * up + p0 + p1])
    wk = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)
    out = F.conv2d(out, wk)
    out = out.reshape(-1, 1, h * up + p0 + p1 - kh + 1, w * up + p0 + p1 - kw + 1)
    out = out.permute(0, 2, 3, 1)
    out = out[:, ::down, ::down, :]
    oh = (h * up + p0 + p1 - kh) // down + 1
    ow = (w * up + p0 + p1 - kw) // down + 1
    return out.view(-1, c, oh, ow)


def _fir_kernel(k=(1, 3, 3, 1), gain=1.0):
    k = torch.tensor(k, dtype=torch.float32)
    k = torch.outer(k, k)
    k /= torch.sum(k)
    return k * gain


def upsample_2d(x, kernel=(1, 3, 3, 1), factor=2, gain=1):
    k = _fir_kernel(kernel, gain * factor ** 2).to(x.device)
    pv = k.shape[0] - factor
    return upfirdn2d_native(x, k, up=factor, pad=((pv + 1) // 2 + factor - 1, pv // 2))


def downsample_2d(x, kernel=(1, 3, 3, 1), factor=2, gain=1):
    k = _fir_kernel(kernel, gain).to(x.device)
    pv = k.shape[0] - factor
    return upfirdn2d_native(x, k, down=factor, pad=((pv + 1) // 2, pv // 2))


# ------------------------------------------------------------------------------------------------ blocks
class GaussianFourierProjection(nn.Module):
    """[UPSTREAM] log=True, set_W_to_weight=True, flip_sin_to_cos=False as UNet2DModel builds it."""

    def __init__(self, embedding_size: int, scale: float = 16.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def forward(self, x):
        x = torch.log(x)
        xp = x[:, None] * self.weight[None, :] * 2 * np.pi
        return torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)


class ResnetBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, eps, groups, groups_out=None, scale=SQRT2, up=False, down=False, use_in_shortcut=None):
        super().__init__()
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups_out, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.up, self.down, self.scale = up, down, scale
        sc = (cin != cout) if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if sc else None

    def forward(self, x, temb):
        h = F.silu(self.norm1(x))
        if self.up:
            x, h = upsample_2d(x), upsample_2d(h)
        elif self.down:
            x, h = downsample_2d(x), downsample_2d(h)
        h = self.conv1(h)
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.scale


class AttentionBlock(nn.Module):
    """[UPSTREAM] AttentionBlock (single head when num_head_channels is None): (attn(x) + x) / rescale_output_factor."""

    def __init__(self, ch, eps, groups=32, head_dim=None, rescale=SQRT2):
        super().__init__()
        self.heads = 1 if head_dim is None else ch // head_dim
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = _ToOut([nn.Linear(ch, ch)])
        self.rescale = rescale

    def forward(self, x):
        b, c, hh, ww = x.shape
        h = self.group_norm(x).view(b, c, hh * ww).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        nh, d = self.heads, c // self.heads
        sp = lambda z: z.view(b, -1, nh, d).permute(0, 2, 1, 3).reshape(b * nh, -1, d)
        q, k, v = sp(q), sp(k), sp(v)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(d)), dim=-1)
        h = torch.bmm(p, v).view(b, nh, -1, d).permute(0, 2, 1, 3).reshape(b, -1, c)
        h = self.to_out[0](h).transpose(1, 2).reshape(b, c, hh, ww)
        return (h + x) / self.rescale


def _g(ch):
    return min(ch // 4, 32)


class _FirDown(nn.Module):
    def forward(self, x):
        return downsample_2d(x)


class _FirUp(nn.Module):
    def forward(self, x):
        return upsample_2d(x)


class SkipDownBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, n_layers, eps, attn, head_dim, add_down, img_ch=3):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, temb_dim, eps, _g(cin if i == 0 else cout), _g(cout))
                                      for i in range(n_layers)])
        self.attentions = nn.ModuleList([AttentionBlock(cout, eps, 32, head_dim) for _ in range(n_layers)]) if attn else None
        if add_down:
            self.resnet_down = ResnetBlock(cout, cout, temb_dim, eps, _g(cout), down=True, use_in_shortcut=True)
            self.downsamplers = nn.ModuleList([_FirDown()])
            self.skip_conv = nn.Conv2d(img_ch, cout, 1)
        else:
            self.resnet_down = self.downsamplers = self.skip_conv = None

    def forward(self, h, temb, skip):
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.resnet_down(h, temb)
            skip = self.downsamplers[0](skip)
            h = self.skip_conv(skip) + h
            outs.append(h)
        return h, outs, skip


class SkipUpBlock(nn.Module):
    def __init__(self, cin, prev, cout, temb_dim, n_layers, eps, attn, head_dim, add_up, img_ch=3):
        super().__init__()
        res = []
        for i in range(n_layers):
            skip_ch = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock(rin + skip_ch, cout, temb_dim, eps, _g(rin + skip_ch), _g(cout)))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([AttentionBlock(cout, eps, 32, head_dim)]) if attn else None
        self.upsampler = _FirUp()
        if add_up:
            self.resnet_up = ResnetBlock(cout, cout, temb_dim, eps, _g(cout), _g(cout), up=True, use_in_shortcut=True)
            self.skip_conv = nn.Conv2d(cout, img_ch, 3, padding=1)
            self.skip_norm = nn.GroupNorm(_g(cout), cout, eps=eps)
        else:
            self.resnet_up = self.skip_conv = self.skip_norm = None

    def forward(self, h, skips: List[torch.Tensor], temb, skip_sample):
        for r in self.resnets:
            h = r(torch.cat([h, skips.pop()], dim=1), temb)
        if self.attentions is not None:
            h = self.attentions[0](h)
        skip_sample = self.upsampler(skip_sample) if skip_sample is not None else 0
        if self.resnet_up is not None:
            skip_sample = skip_sample + self.skip_conv(F.silu(self.skip_norm(h)))
            h = self.resnet_up(h, temb)
        return h, skip_sample


class _Mid(nn.Module):
    def __init__(self, ch, temb_dim, eps, head_dim, scale):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(ch, ch, temb_dim, eps, _g(ch), scale=scale) for _ in range(2)])
        self.attentions = nn.ModuleList([AttentionBlock(ch, eps, _g(ch), head_dim, rescale=scale)])

    def forward(self, h, temb):
        return self.resnets[1](self.attentions[0](self.resnets[0](h, temb)), temb)


class NCSNppRef(nn.Module):
    """[UPSTREAM] UNet2DModel(time_embedding_type="fourier", Skip blocks, norm_num_groups=None), model.py:839-857."""

    def __init__(self, in_channels=3, out_channels=3, sample_size=32, block_out_channels: Sequence[int] = (128, 256, 256, 256),
                 down_block_types=("SkipDownBlock2D", "AttnSkipDownBlock2D", "SkipDownBlock2D", "SkipDownBlock2D"),
                 up_block_types=("SkipUpBlock2D", "SkipUpBlock2D", "AttnSkipUpBlock2D", "SkipUpBlock2D"), layers_per_block=4,
                 norm_eps=1e-6, attention_head_dim=None, mid_block_scale_factor=SQRT2, **_ignored):
        super().__init__()
        boc = list(block_out_channels)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, sample_size=sample_size,
                                      block_out_channels=tuple(boc), down_block_types=tuple(down_block_types),
                                      up_block_types=tuple(up_block_types), layers_per_block=layers_per_block)
        temb = boc[0] * 4
        self.time_proj = GaussianFourierProjection(boc[0], 16.0)
        self.time_embedding = TimestepEmbedding(2 * boc[0], temb)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        downs, ch = [], boc[0]
        for i, typ in enumerate(down_block_types):
            downs.append(SkipDownBlock(ch, boc[i], temb, layers_per_block, norm_eps, typ.startswith("Attn"), attention_head_dim,
                                       i != len(boc) - 1, in_channels))
            ch = boc[i]
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _Mid(ch, temb, norm_eps, attention_head_dim, mid_block_scale_factor)
        rev = boc[::-1]
        ups, out_ch = [], rev[0]
        for i, typ in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            ups.append(SkipUpBlock(in_ch, prev, out_ch, temb, layers_per_block + 1, norm_eps, typ.startswith("Attn"), attention_head_dim,
                                   i != len(boc) - 1, out_channels))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(_g(boc[0]), boc[0], eps=norm_eps)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, return_dict: bool = False):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32)
        if t.dim() == 0:
            t = t[None]
        t = t.to(torch.float32) * torch.ones(sample.shape[0], dtype=torch.float32)
        emb = self.time_embedding(self.time_proj(t))
        skip = sample
        h = self.conv_in(sample)
        res = [h]
        for blk in self.down_blocks:
            h, outs, skip = blk(h, emb, skip)
            res.extend(outs)
        h = self.mid_block(h, emb)
        skip = None
        for blk in self.up_blocks:
            n = len(blk.resnets)
            mine, res = res[-n:], res[:-n]
            h, skip = blk(h, mine, emb, skip)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        if skip is not None:
            h = h + skip
        h = h / t.reshape(-1, 1, 1, 1)
        return SimpleNamespace(sample=h) if return_dict else (h,)
