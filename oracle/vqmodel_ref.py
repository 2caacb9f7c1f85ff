"""CPU oracle: the VQ-VAE (``VQModel``) the reference's latent-diffusion path decodes with (model.py:713, loss.py:942-962,
VillanDiffusion.py:378,472; config of ``CompVis/ldm-celebahq-256``/vqvae).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in the un-vendored diffusers fork
(requirement.txt:37); this restates the published upstream module (diffusers ~0.16 ``VQModel`` = ``Encoder`` +
``quant_conv`` + ``VectorQuantizer`` + ``post_quant_conv`` + ``Decoder``) with diffusers state-dict names.  Checked by
known-answer tests in tests/test_vqmodel_oracle.py (parameter count of the published config, nearest-code property of the
quantiser, shapes, encode/decode call contract).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import Attention, Downsample2D, Upsample2D


class ResnetBlockNoTemb(nn.Module):
    """[UPSTREAM] ResnetBlock2D(temb_channels=None)."""

    def __init__(self, cin: int, cout: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class _Mid(nn.Module):
    def __init__(self, ch, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlockNoTemb(ch, ch, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, groups, eps, None)])

    def forward(self, h):
        return self.resnets[1](self.attentions[0](self.resnets[0](h)))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, n_layers, groups, eps, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlockNoTemb(cin if i == 0 else cout, cout, groups, eps) for i in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 0)]) if add_down else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        return self.downsamplers[0](h) if self.downsamplers is not None else h


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, n_layers, groups, eps, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlockNoTemb(cin if i == 0 else cout, cout, groups, eps) for i in range(n_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        return self.upsamplers[0](h) if self.upsamplers is not None else h


class Encoder(nn.Module):
    def __init__(self, cin, cout, boc: Sequence[int], layers, groups, eps):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, oc in enumerate(boc):
            blocks.append(_EncBlock(ch, oc, layers, groups, eps, add_down=i != len(boc) - 1))
            ch = oc
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid(ch, groups, eps)
        self.conv_norm_out = nn.GroupNorm(groups, ch, eps=eps)
        self.conv_out = nn.Conv2d(ch, cout, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.down_blocks:
            h = b(h)
        h = self.mid_block(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, cin, cout, boc: Sequence[int], layers, groups, eps):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(cin, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups, eps)
        blocks, ch = [], rev[0]
        for i, oc in enumerate(rev):
            blocks.append(_DecBlock(ch, oc, layers + 1, groups, eps, add_up=i != len(rev) - 1))
            ch = oc
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, ch, eps=eps)
        self.conv_out = nn.Conv2d(ch, cout, 3, padding=1)

    def forward(self, z):
        h = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            h = b(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class VectorQuantizer(nn.Module):
    """[UPSTREAM] VectorQuantizer (remap=None, sane_index_shape=False, legacy=True): nearest code in L2."""

    def __init__(self, n_e: int, e_dim: int, beta: float = 0.25):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def forward(self, z):
        zp = z.permute(0, 2, 3, 1).contiguous()
        zf = zp.view(-1, self.e_dim)
        w = self.embedding.weight
        d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1) - 2 * zf @ w.t()
        idx = torch.argmin(d, dim=1)
        zq = self.embedding(idx).view(zp.shape)
        zq = zp + (zq - zp).detach()
        return zq.permute(0, 3, 1, 2).contiguous(), idx


class VQModelRef(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512), layers_per_block=2, latent_channels=3,
                 num_vq_embeddings=8192, norm_num_groups=32, vq_embed_dim=None, sample_size=256, scaling_factor=0.18215,
                 norm_eps=1e-6, **_ignored):
        super().__init__()
        vq_embed_dim = vq_embed_dim if vq_embed_dim is not None else latent_channels
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, latent_channels=latent_channels,
                                      num_vq_embeddings=num_vq_embeddings, norm_num_groups=norm_num_groups,
                                      vq_embed_dim=vq_embed_dim, sample_size=sample_size, scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups, norm_eps)
        self.quant_conv = nn.Conv2d(latent_channels, vq_embed_dim, 1)
        self.quantize = VectorQuantizer(num_vq_embeddings, vq_embed_dim)
        self.post_quant_conv = nn.Conv2d(vq_embed_dim, latent_channels, 1)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, norm_eps)

    def encode(self, x):
        """Latents are NOT quantised at encode time (upstream VQModel.encode)."""
        return SimpleNamespace(latents=self.quant_conv(self.encoder(x)))

    def decode(self, h, force_not_quantize: bool = False):
        quant = h if force_not_quantize else self.quantize(h)[0]
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(quant)))

    def forward(self, x):
        return self.decode(self.encode(x).latents)
