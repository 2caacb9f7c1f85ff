"""CPU oracle: upstream-faithful restatement of diffusers' ``UNet2DModel``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the class lives
in the un-vendored diffusers fork (requirement.txt:37); the architecture is
pinned in-tree only by the config at reference model.py:816-834 and the forward
pass by SURVEY.md §3.4.  Checked by: parameter count 35 746 307, state-dict key
list and fp64 gradient checks in tests/.

Everything here is plain ``torch.nn.functional`` on whatever device/dtype the
module is moved to (the tests use CPU fp32 and CPU fp64).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# reference model.py:816-834 (the from-scratch DDPM-CIFAR10 config)
DDPM_CIFAR10_CONFIG = dict(
    in_channels=3,
    out_channels=3,
    sample_size=32,
    block_out_channels=(128, 256, 256, 256),
    down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
    layers_per_block=2,
    norm_num_groups=32,
    norm_eps=1e-6,
    downsample_padding=0,
    flip_sin_to_cos=False,
    freq_shift=1,
    attention_head_dim=None,
)


def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float,
                       max_period: int = 10000) -> torch.Tensor:
    """[UPSTREAM] diffusers ``get_timestep_embedding`` (scale=1)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = t[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, temb_dim: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class _ToOut(nn.ModuleList):
    """``to_out.0`` naming of the >=0.17 attention block."""


class Attention(nn.Module):
    """Single-head spatial self-attention ([UPSTREAM] ``AttentionBlock``)."""

    def __init__(self, ch: int, groups: int, eps: float, head_dim: Optional[int]):
        super().__init__()
        self.heads = 1 if head_dim is None else ch // head_dim
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = _ToOut([nn.Linear(ch, ch)])

    def forward(self, x):
        b, c, hgt, wid = x.shape
        h = self.group_norm(x).view(b, c, hgt * wid).transpose(1, 2)  # [B, HW, C]
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        nh, d = self.heads, c // self.heads

        def split(z):
            return z.view(b, -1, nh, d).permute(0, 2, 1, 3).reshape(b * nh, -1, d)

        q, k, v = split(q), split(k), split(v)
        scores = torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(d))
        probs = torch.softmax(scores.float(), dim=-1).to(scores.dtype)
        h = torch.bmm(probs, v).view(b, nh, -1, d).permute(0, 2, 1, 3).reshape(b, -1, c)
        h = self.to_out[0](h).transpose(1, 2).reshape(b, c, hgt, wid)
        return h + x


class Downsample2D(nn.Module):
    def __init__(self, ch: int, padding: int):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), value=0.0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, n_layers, groups, eps, attn, head_dim, add_down, ds_pad):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, groups, eps) for i in range(n_layers)])
        if attn:
            self.attentions = nn.ModuleList([Attention(cout, groups, eps, head_dim) for _ in range(n_layers)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout, ds_pad)]) if add_down else None

    def forward(self, h, temb):
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb_dim, groups, eps, head_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_dim, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, groups, eps, head_dim)])

    def forward(self, h, temb):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h)
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, prev, cout, temb_dim, n_layers, groups, eps, attn, head_dim, add_up):
        super().__init__()
        res = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb_dim, groups, eps))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([Attention(cout, groups, eps, head_dim) for _ in range(n_layers)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h, skips: List[torch.Tensor], temb):
        for i, r in enumerate(self.resnets):
            h = r(torch.cat([h, skips.pop()], dim=1), temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class UNet2DModelRef(nn.Module):
    """[UPSTREAM] ``UNet2DModel`` with positional time embedding, diffusers state-dict names."""

    def __init__(self, in_channels=3, out_channels=3, sample_size=32,
                 block_out_channels: Sequence[int] = (128, 256, 256, 256),
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 layers_per_block=2, norm_num_groups=32, norm_eps=1e-6, downsample_padding=0,
                 flip_sin_to_cos=False, freq_shift=1, attention_head_dim=None, **_ignored):
        super().__init__()
        self.in_channels, self.out_channels, self.sample_size = in_channels, out_channels, sample_size
        self.flip_sin_to_cos, self.freq_shift = flip_sin_to_cos, freq_shift
        boc = list(block_out_channels)
        self.time_dim0 = boc[0]
        temb_dim = boc[0] * 4
        g, eps, hd = norm_num_groups, norm_eps, attention_head_dim
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_dim)
        downs, out_ch = [], boc[0]
        for i, typ in enumerate(down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            downs.append(DownBlock(in_ch, out_ch, temb_dim, layers_per_block, g, eps,
                                   typ == "AttnDownBlock2D", hd, i != len(boc) - 1, downsample_padding))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb_dim, g, eps, hd)
        rev = boc[::-1]
        ups, out_ch = [], rev[0]
        for i, typ in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            ups.append(UpBlock(in_ch, prev, out_ch, temb_dim, layers_per_block + 1, g, eps,
                               typ == "AttnUpBlock2D", hd, i != len(boc) - 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, sample: torch.Tensor, timestep, return_dict: bool = False):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        if t.dim() == 0:
            t = t[None]
        t = t.to(sample.device) * torch.ones(sample.shape[0], dtype=t.dtype, device=sample.device)
        temb = timestep_embedding(t, self.time_dim0, self.flip_sin_to_cos, self.freq_shift).to(sample.dtype)
        temb = self.time_embedding(temb)
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, temb)
            skips.extend(outs)
        h = self.mid_block(h, temb)
        for blk in self.up_blocks:
            h = blk(h, skips, temb)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return (h,)

    # diffusers <0.17 checkpoints name the attention projections query/key/value/proj_attn
    LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}

    @classmethod
    def convert_legacy_keys(cls, sd: dict) -> dict:
        out = {}
        for k, v in sd.items():
            parts = k.split(".")
            if "attentions" in parts and len(parts) >= 2 and parts[-2] in cls.LEGACY_ATTN:
                parts[-2] = cls.LEGACY_ATTN[parts[-2]]
                k = ".".join(parts)
            out[k] = v
        return out
