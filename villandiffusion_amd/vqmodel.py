"""VQ-VAE (``VQModel``) of the latent-diffusion path, forward only, on the HIP kernels of the UNet.

Reference use: ``vae.encode(x).latents`` / ``vae.decode(z).sample`` in loss.py:942-962,
VillanDiffusion.py:378,472 and inside the LDM pipeline (model.py:713); the reference freezes it
(``vae.requires_grad_(False)``, model.py:790), so there is no backward here.  Architecture = diffusers ``VQModel``
(Encoder -> quant_conv -> VectorQuantizer -> post_quant_conv -> Decoder), state-dict names as in diffusers so the
``vqvae/`` folder of ``CompVis/ldm-celebahq-256`` loads unchanged (legacy attention key names are mapped).

Every tensor op is a launch through the C ABI (3x3 convs incl. the fused nearest-2x upsample and the padded stride-2
downsample, GroupNorm+SiLU, 1x1 convs, attention GEMMs + column softmax, ``vd_vq_nearest``); parameters are views of one
flat fp32 buffer like the UNet's.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .lib import B_CONV3, B_CONV3_S2, B_CONV3_UP
from .unet import LEGACY_ATTN, _Attn, _Conv, _Norm, _ensure_path


class _ResnetNoTemb:
    """ResnetBlock2D(temb_channels=None): conv2(silu(gn(conv1(silu(gn(x)))))) + shortcut(x)."""

    def __init__(self, net, prefix, cin, cout):
        self.net, self.prefix, self.cin, self.cout = net, prefix, cin, cout
        self.norm1 = _Norm(net, prefix + ".norm1", cin, True)
        self.conv1 = _Conv(net, prefix + ".conv1", cin, cout)
        self.norm2 = _Norm(net, prefix + ".norm2", cout, True)
        self.conv2 = _Conv(net, prefix + ".conv2", cout, cout)
        self.has_sc = cin != cout
        if self.has_sc:
            net._decl(prefix + ".conv_shortcut.weight", (cout, cin, 1, 1), fan_in=cin)
            net._decl(prefix + ".conv_shortcut.bias", (cout,), fan_in=cin, is_bias=True)

    def fwd(self, x):
        net = self.net
        B, _, H, W = x.shape
        a1 = torch.empty_like(x)
        self.norm1.fwd(x, a1)
        h1 = torch.empty((B, self.cout, H, W), device=x.device, dtype=torch.float32)
        self.conv1.fwd(a1, h1)
        del a1
        a2 = torch.empty_like(h1)
        self.norm2.fwd(h1, a2)
        out = h1                                     # conv2 does not read h1: reuse its storage for the block output
        if self.has_sc:
            ops.conv1x1(x, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                        net.P[self.prefix + ".conv_shortcut.bias"], out)
            self.conv2.fwd(a2, out, residual=out)
        else:
            self.conv2.fwd(a2, out, residual=x)
        return out


class _Mid:
    def __init__(self, net, prefix, ch):
        self.r0 = _ResnetNoTemb(net, prefix + ".resnets.0", ch, ch)
        self.attn = _Attn(net, prefix + ".attentions.0", ch, None)
        self.r1 = _ResnetNoTemb(net, prefix + ".resnets.1", ch, ch)

    def fwd(self, h):
        h = self.r0.fwd(h)
        out = torch.empty_like(h)
        self.attn.fwd(h, out, None, False)
        return self.r1.fwd(out)


class VQModel(nn.Module):
    """Drop-in for diffusers ``VQModel`` on the inference surface the reference uses: ``.encode(x).latents``,
    ``.decode(z).sample``, ``.config``, ``.device``, ``.eval()``, ``.requires_grad_``, ``state_dict``/``load_state_dict``."""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 3,
                 up_block_types=("UpDecoderBlock2D",) * 3, block_out_channels=(128, 256, 512), layers_per_block=2, act_fn="silu",
                 latent_channels=3, sample_size=256, num_vq_embeddings=8192, norm_num_groups=32, vq_embed_dim=None,
                 scaling_factor=0.18215, norm_eps=1e-6, device=None, **unused):
        super().__init__()
        if act_fn != "silu" or any(t != "DownEncoderBlock2D" for t in down_block_types) or \
                any(t != "UpDecoderBlock2D" for t in up_block_types):
            raise NotImplementedError("only the published VQModel configuration family (silu, Down/UpDecoderBlock2D) is implemented")
        vq_embed_dim = vq_embed_dim if vq_embed_dim is not None else latent_channels
        boc = tuple(block_out_channels)
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, down_block_types=tuple(down_block_types),
            up_block_types=tuple(up_block_types), block_out_channels=boc, layers_per_block=layers_per_block, act_fn=act_fn,
            latent_channels=latent_channels, sample_size=sample_size, num_vq_embeddings=num_vq_embeddings,
            norm_num_groups=norm_num_groups, vq_embed_dim=vq_embed_dim, scaling_factor=scaling_factor)
        self.groups, self.eps = norm_num_groups, norm_eps
        if device is not None:
            self._dev = torch.device(device)
        elif torch.cuda.is_available():
            self._dev = torch.device("cuda", torch.cuda.current_device())
        else:
            self._dev = torch.device("cpu")       # structure-only use; compute fails loudly in lib.require_device()
        self._decls: List[Tuple[str, Tuple[int, ...], dict]] = []
        self._qkv: List[Tuple[str, int]] = []

        # ---- encoder ----
        self.e_in = _Conv(self, "encoder.conv_in", in_channels, boc[0])
        self.e_blocks, ch = [], boc[0]
        for i, oc in enumerate(boc):
            res = [_ResnetNoTemb(self, f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else oc, oc) for j in range(layers_per_block)]
            ds = _Conv(self, f"encoder.down_blocks.{i}.downsamplers.0.conv", oc, oc, mode=B_CONV3_S2) if i != len(boc) - 1 else None
            self.e_blocks.append((res, ds))
            ch = oc
        self.e_mid = _Mid(self, "encoder.mid_block", ch)
        self.e_norm = _Norm(self, "encoder.conv_norm_out", ch, True)
        self.e_out = _Conv(self, "encoder.conv_out", ch, latent_channels)
        # ---- quantiser ----
        self._decl("quant_conv.weight", (vq_embed_dim, latent_channels, 1, 1), fan_in=latent_channels)
        self._decl("quant_conv.bias", (vq_embed_dim,), fan_in=latent_channels, is_bias=True)
        self._decl("quantize.embedding.weight", (num_vq_embeddings, vq_embed_dim), codebook=True)
        self._decl("post_quant_conv.weight", (latent_channels, vq_embed_dim, 1, 1), fan_in=vq_embed_dim)
        self._decl("post_quant_conv.bias", (latent_channels,), fan_in=vq_embed_dim, is_bias=True)
        # ---- decoder ----
        rev = list(reversed(boc))
        self.d_in = _Conv(self, "decoder.conv_in", latent_channels, rev[0])
        self.d_mid = _Mid(self, "decoder.mid_block", rev[0])
        self.d_blocks, ch = [], rev[0]
        for i, oc in enumerate(rev):
            res = [_ResnetNoTemb(self, f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else oc, oc) for j in range(layers_per_block + 1)]
            us = _Conv(self, f"decoder.up_blocks.{i}.upsamplers.0.conv", oc, oc, mode=B_CONV3_UP) if i != len(rev) - 1 else None
            self.d_blocks.append((res, us))
            ch = oc
        self.d_norm = _Norm(self, "decoder.conv_norm_out", ch, True)
        self.d_out = _Conv(self, "decoder.conv_out", ch, out_channels)
        self._materialise()

    # ------------------------------------------------------------------------------------------ parameter plumbing
    def _decl(self, name, shape, fan_in=None, is_bias=False, ones=False, zeros=False, codebook=False):
        self._decls.append((name, tuple(shape), dict(fan_in=fan_in, is_bias=is_bias, ones=ones, zeros=zeros, codebook=codebook)))

    def _decl_qkv(self, prefix, ch):
        self._qkv.append((prefix, ch))
        return prefix + "::qkv_w", prefix + "::qkv_b"

    def _materialise(self):
        layout = []
        for prefix, ch in self._qkv:                 # q, k, v adjacent: one [3C, C] projection per attention block
            for n in ("to_q", "to_k", "to_v"):
                layout.append((f"{prefix}.{n}.weight", (ch, ch), dict(fan_in=ch)))
            for n in ("to_q", "to_k", "to_v"):
                layout.append((f"{prefix}.{n}.bias", (ch,), dict(fan_in=ch, is_bias=True)))
        layout.extend(self._decls)
        offs, total = {}, 0
        for name, shape, _ in layout:
            n = int(math.prod(shape))
            offs[name] = (total, n, shape)
            total += (n + 3) // 4 * 4
        self._layout, self._offs, self.flat_numel = layout, offs, total
        self.flat_param = torch.zeros(total, device=self._dev, dtype=torch.float32)
        self.P: Dict[str, torch.Tensor] = {}
        for name, shape, _ in layout:
            off, n, _ = offs[name]
            parts = name.split(".")
            holder = _ensure_path(self, parts[:-1])
            p = nn.Parameter(self.flat_param[off:off + n].view(shape), requires_grad=False)
            holder.register_parameter(parts[-1], p)
            self.P[name] = p.data
        self.Pq = {}
        for prefix, ch in self._qkv:
            ow, ob = offs[f"{prefix}.to_q.weight"][0], offs[f"{prefix}.to_q.bias"][0]
            self.Pq[prefix + "::qkv_w"] = self.flat_param[ow:ow + 3 * ch * ch].view(3 * ch, ch)
            self.Pq[prefix + "::qkv_b"] = self.flat_param[ob:ob + 3 * ch]
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None):
        """torch default init of Conv2d / Linear / GroupNorm; codebook U(-1/n_e, 1/n_e) like upstream VectorQuantizer."""
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        host = torch.zeros(self.flat_numel, dtype=torch.float32)
        for name, shape, meta in self._layout:
            off, n, _ = self._offs[name]
            if meta.get("ones"):
                host[off:off + n] = 1.0
            elif meta.get("zeros"):
                host[off:off + n] = 0.0
            else:
                bound = 1.0 / shape[0] if meta.get("codebook") else 1.0 / math.sqrt(meta["fan_in"])
                host[off:off + n] = (torch.rand(n, generator=gen) * 2 - 1) * bound
        self.flat_param.copy_(host)

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {}
        for k, v in state_dict.items():
            parts = k.split(".")
            if "attentions" in parts and len(parts) >= 2 and parts[-2] in LEGACY_ATTN:
                parts[-2] = LEGACY_ATTN[parts[-2]]
                k = ".".join(parts)
            sd[k] = v
        missing = [k for k in self._offs if k not in sd]
        unexpected = [k for k in sd if k not in self._offs]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        with torch.no_grad():
            for k, v in sd.items():
                if k in self._offs:
                    off, n, shape = self._offs[k]
                    assert v.numel() == n, (k, v.shape, shape)
                    self.flat_param[off:off + n].copy_(v.reshape(-1).to(torch.float32))
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def to(self, *args, **kwargs):
        return self

    def cuda(self, device=None):
        return self

    @property
    def device(self):
        return self._dev

    @property
    def dtype(self):
        return torch.float32

    # ------------------------------------------------------------------------------------------ forward launch sequences
    def _conv1x1(self, name, x):
        w = self.P[name + ".weight"]
        out = torch.empty((x.shape[0], w.shape[0], x.shape[2], x.shape[3]), device=x.device, dtype=torch.float32)
        return ops.conv1x1(x, w.view(w.shape[0], w.shape[1]), self.P[name + ".bias"], out)

    @staticmethod
    def _new(x, ch, scale=1.0):
        B, _, H, W = x.shape
        return torch.empty((B, ch, int(H * scale), int(W * scale)), device=x.device, dtype=torch.float32)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """latents = quant_conv(Encoder(x)); NOT quantised (upstream VQModel.encode)."""
        x = x.to(self._dev, torch.float32).contiguous()
        h = self._new(x, self.e_in.cout)
        self.e_in.fwd(x, h)
        for res, ds in self.e_blocks:
            for r in res:
                h = r.fwd(h)
            if ds is not None:
                o = self._new(h, ds.cout, 0.5)
                ds.fwd(h, o)
                h = o
        h = self.e_mid.fwd(h)
        a = torch.empty_like(h)
        self.e_norm.fwd(h, a)
        z = self._new(a, self.e_out.cout)
        self.e_out.fwd(a, z)
        lat = self._conv1x1("quant_conv", z)
        return SimpleNamespace(latents=lat) if return_dict else (lat,)

    @torch.no_grad()
    def quantize_latents(self, h: torch.Tensor, return_indices: bool = False):
        h = h.to(self._dev, torch.float32).contiguous()
        zq = torch.empty_like(h)
        idx = torch.empty(h.shape[0] * h.shape[2] * h.shape[3], device=h.device, dtype=torch.int64) if return_indices else None
        ops.vq_nearest(h, self.P["quantize.embedding.weight"], zq, idx)
        return (zq, idx) if return_indices else zq

    @torch.no_grad()
    def decode(self, h: torch.Tensor, force_not_quantize: bool = False, return_dict: bool = True):
        """sample = Decoder(post_quant_conv(quantize(h)))  (upstream VQModel.decode)."""
        h = h.to(self._dev, torch.float32).contiguous()
        q = h if force_not_quantize else self.quantize_latents(h)
        q = self._conv1x1("post_quant_conv", q)
        x = self._new(q, self.d_in.cout)
        self.d_in.fwd(q, x)
        x = self.d_mid.fwd(x)
        for res, us in self.d_blocks:
            for r in res:
                x = r.fwd(x)
            if us is not None:
                o = self._new(x, us.cout, 2.0)
                us.fwd(x, o)
                x = o
        a = torch.empty_like(x)
        self.d_norm.fwd(x, a)
        out = self._new(a, self.d_out.cout)
        self.d_out.fwd(a, out)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def forward(self, x, return_dict: bool = True):
        return self.decode(self.encode(x).latents, return_dict=return_dict)
