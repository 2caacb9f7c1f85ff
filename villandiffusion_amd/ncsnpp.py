"""NCSN++ score network for SDE-VE: diffusers ``UNet2DModel`` with ``SkipDownBlock2D /
AttnSkipDownBlock2D / SkipUpBlock2D / AttnSkipUpBlock2D``, Gaussian-Fourier time embedding, FIR (1,3,3,1) resampling,
``norm_num_groups=None`` (groups = min(ch // 4, 32)), residual / attention outputs divided by sqrt(2), an input-image
pyramid added on the way down and an output-image pyramid accumulated on the way up, and the final division by sigma --
the network the reference builds in model.py:839-857 / 876-894 (``NCSNPP-*`` ids, ``fusing/cifar10-ncsnpp-ve``,
``google/ncsnpp-celebahq-256``).  State-dict names are diffusers' (legacy attention keys are mapped).

Same construction as ``unet.UNet2DModel`` (whose parameter plumbing it inherits): one flat fp32 parameter / gradient
buffer, all ``time_emb_proj`` matrices fused into one GEMM, explicit forward AND backward launch sequences through the
C ABI wrapped in one ``autograd.Function``.  Skip concatenations are two strided-copy launches into the concat buffer.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .lib import A_COL, B_CONV3, B_PLAIN
from .unet import UNet2DModel, _Attn, _Conv, _Norm, _bx3_packed_1x1, _wgrad1x1_math

SQRT2 = math.sqrt(2.0)


def _g(ch: int) -> int:
    return min(ch // 4, 32)


class _ResnetPP:
    """ResnetBlock2D(output_scale_factor=s, up/down='fir', use_in_shortcut):  y = (shortcut(R(x)) + conv2(silu(gn2(conv1(R(silu(gn1(x))))
    + temb))) / s   with R = FIR up / down / identity."""

    def __init__(self, net, prefix, cin, cout, groups, groups_out=None, scale=SQRT2, up=False, down=False, use_in_shortcut=None):
        self.net, self.prefix, self.cin, self.cout = net, prefix, cin, cout
        self.scale, self.up, self.down = scale, up, down
        self.norm1 = _Norm(net, prefix + ".norm1", cin, True, groups)
        self.conv1 = _Conv(net, prefix + ".conv1", cin, cout)
        self.temb_off = net._decl_temb(prefix + ".time_emb_proj", cout)
        self.norm2 = _Norm(net, prefix + ".norm2", cout, True, groups if groups_out is None else groups_out)
        self.conv2 = _Conv(net, prefix + ".conv2", cout, cout)
        self.has_sc = (cin != cout) if use_in_shortcut is None else use_in_shortcut
        if self.has_sc:
            net._decl(prefix + ".conv_shortcut.weight", (cout, cin, 1, 1), fan_in=cin)
            net._decl(prefix + ".conv_shortcut.bias", (cout,), fan_in=cin, is_bias=True)

    def _resample(self, t):
        B, C, H, W = t.shape
        out = torch.empty((B, C, 2 * H, 2 * W) if self.up else (B, C, H // 2, W // 2), device=t.device, dtype=torch.float32)
        return ops.fir_resample2(t, out, up=self.up)

    def fwd(self, x, st, save):
        net = self.net
        B, _, H, W = x.shape
        a1 = torch.empty_like(x)
        m1, r1 = self.norm1.fwd(x, a1)
        xs, a1s = x, a1
        if self.up or self.down:
            xs, a1s = self._resample(x), self._resample(a1)
        _, _, H2, W2 = a1s.shape
        h1 = torch.empty((B, self.cout, H2, W2), device=x.device, dtype=torch.float32)
        self.conv1.fwd(a1s, h1, rowadd=st.temb_all[:, self.temb_off:], rowadd_bstride=st.temb_all.stride(0))
        a2 = torch.empty_like(h1)
        m2, r2 = self.norm2.fwd(h1, a2)
        out = torch.empty_like(h1)
        if self.has_sc:
            ops.conv1x1(xs, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                        net.P[self.prefix + ".conv_shortcut.bias"], out,
                        a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", False, self.cout, self.cin,
                                                 xs.shape[2] * xs.shape[3], xs.shape[0]))
            self.conv2.fwd(a2, out, residual=out)
        else:
            self.conv2.fwd(a2, out, residual=xs)
        ops.scale_(out, 1.0 / self.scale)
        return out, ((x, a1, m1, r1, xs, a1s, h1, a2, m2, r2) if save else None)

    def bwd(self, saved, dout, st):
        """returns dx (new tensor)."""
        net = self.net
        x, a1, m1, r1, xs, a1s, h1, a2, m2, r2 = saved
        B = x.shape[0]
        dev = x.device
        dsum = torch.empty_like(dout)
        ops.lincomb(dsum, [dout.contiguous()], [1.0 / self.scale])           # gradient wrt (shortcut + h)
        bias_ws = net.scratch_bc(B, self.cout)
        ops.rowsum(dsum, bias_ws)
        da2 = torch.empty_like(a2)
        self.conv2.bwd(dsum, a2, da2, bias_ws=bias_ws)
        dh1 = torch.empty_like(h1)
        self.norm2.bwd(da2, h1, m2, r2, dh1)
        dt = st.d_temb_all[:, self.temb_off:self.temb_off + self.cout]
        ops.rowsum(dh1, dt, ws_ld=st.d_temb_all.stride(0))
        da1s = torch.empty_like(a1s)
        self.conv1.bwd(dh1, a1s, da1s, bias_ws=dt)
        # shortcut branch gradient wrt xs
        if self.has_sc:
            wsc = net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin)
            ops.conv_wgrad(dsum, xs, net.G[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin), B_PLAIN, net.wgrad_ws,
                           accumulate=True, math_mode=_wgrad1x1_math(net, dsum, xs))
            net.colsum_later(bias_ws, net.G[self.prefix + ".conv_shortcut.bias"], B, self.cout)
            dxs = torch.empty_like(xs)
            HW = xs.shape[2] * xs.shape[3]
            ops.gemm(wsc, dsum, dxs, M=self.cin, N=B * HW, K=self.cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=self.cin, ldb=HW,
                     b_bstride=self.cout * HW, ldd=HW, d_bstride=self.cin * HW,
                     a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", True, self.cin, self.cout, HW, B))
        else:
            dxs = dsum
        if self.up or self.down:
            # adjoint of the FIR resampling: d(up)^T g = 4 down(g);  d(down)^T g = up(g) / 4
            da1 = torch.empty_like(a1)
            dxr = torch.empty_like(x)
            ops.fir_resample2(da1s, da1, up=not self.up, scale=4.0 if self.up else 0.25)
            ops.fir_resample2(dxs, dxr, up=not self.up, scale=4.0 if self.up else 0.25)
        else:
            da1, dxr = da1s, dxs
        dx = torch.empty_like(x)
        self.norm1.bwd(da1, x, m1, r1, dx, extra=dxr)
        return dx


class _AttnPP:
    """AttentionBlock with rescale_output_factor: y = (attn(x) + x) / r."""

    def __init__(self, net, prefix, ch, head_dim, groups, rescale=SQRT2):
        self.attn = _Attn(net, prefix, ch, head_dim, groups)
        self.rescale = rescale

    def fwd(self, x, st, save):
        out = torch.empty_like(x)
        s = self.attn.fwd(x, out, st, save)
        ops.scale_(out, 1.0 / self.rescale)
        return out, s

    def bwd(self, saved, dout, st):
        d = torch.empty_like(dout)
        ops.lincomb(d, [dout.contiguous()], [1.0 / self.rescale])
        dx = torch.empty_like(d)
        self.attn.bwd(saved, d, dx, st)
        return dx


class NCSNppModel(UNet2DModel):
    """Drop-in for ``diffusers.UNet2DModel`` in its NCSN++ configuration (time_embedding_type='fourier', Skip blocks)."""

    def __init__(self, in_channels=3, out_channels=3, sample_size=32, block_out_channels=(128, 256, 256, 256),
                 down_block_types=("SkipDownBlock2D", "AttnSkipDownBlock2D", "SkipDownBlock2D", "SkipDownBlock2D"),
                 up_block_types=("SkipUpBlock2D", "SkipUpBlock2D", "AttnSkipUpBlock2D", "SkipUpBlock2D"), layers_per_block=4,
                 norm_num_groups=None, norm_eps=1e-6, downsample_padding=1, flip_sin_to_cos=True, freq_shift=0,
                 attention_head_dim=None, act_fn="silu", time_embedding_type="fourier", center_input_sample=False,
                 mid_block_scale_factor=1.41421356237, device=None, **unused):
        nn.Module.__init__(self)
        if time_embedding_type != "fourier" or act_fn != "silu" or norm_num_groups is not None or center_input_sample:
            raise NotImplementedError("NCSNppModel implements the NCSN++ configuration of UNet2DModel (fourier embedding, "
                                      "norm_num_groups=None); use unet.UNet2DModel for the DDPM configuration")
        for t in tuple(down_block_types) + tuple(up_block_types):
            if t not in ("SkipDownBlock2D", "AttnSkipDownBlock2D", "SkipUpBlock2D", "AttnSkipUpBlock2D"):
                raise NotImplementedError(f"block type {t}")
        boc = list(block_out_channels)
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, sample_size=sample_size, block_out_channels=tuple(boc),
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types), layers_per_block=layers_per_block,
            norm_num_groups=None, norm_eps=norm_eps, downsample_padding=downsample_padding, flip_sin_to_cos=flip_sin_to_cos,
            freq_shift=freq_shift, attention_head_dim=attention_head_dim, act_fn=act_fn, time_embedding_type=time_embedding_type,
            center_input_sample=center_input_sample, mid_block_scale_factor=mid_block_scale_factor)
        self.in_channels, self.out_channels, self.sample_size = in_channels, out_channels, sample_size
        self.groups, self.eps = 32, norm_eps
        if device is not None:
            self._dev = torch.device(device)
        elif torch.cuda.is_available():
            self._dev = torch.device("cuda", torch.cuda.current_device())
        else:
            self._dev = torch.device("cpu")
        self._decls: List[Tuple[str, Tuple[int, ...], dict]] = []
        self._temb: List[Tuple[str, int]] = []
        self._qkv: List[Tuple[str, int]] = []
        temb_dim = boc[0] * 4
        self.time_dim0, self.temb_dim = 2 * boc[0], temb_dim
        hd = attention_head_dim
        self._decl("time_proj.weight", (boc[0],), fan_in=1, fourier=True)
        self._decl("time_embedding.linear_1.weight", (temb_dim, 2 * boc[0]), fan_in=2 * boc[0])
        self._decl("time_embedding.linear_1.bias", (temb_dim,), fan_in=2 * boc[0], is_bias=True)
        self._decl("time_embedding.linear_2.weight", (temb_dim, temb_dim), fan_in=temb_dim)
        self._decl("time_embedding.linear_2.bias", (temb_dim,), fan_in=temb_dim, is_bias=True)
        self._conv_in = _Conv(self, "conv_in", in_channels, boc[0])
        self.down = []
        ch = boc[0]
        for i, typ in enumerate(down_block_types):
            cin, ch = ch, boc[i]
            blk = SimpleNamespace(res=[], attn=[], down=None, skip_conv=None)
            for j in range(layers_per_block):
                c0 = cin if j == 0 else ch
                blk.res.append(_ResnetPP(self, f"down_blocks.{i}.resnets.{j}", c0, ch, _g(c0), _g(ch)))
                if typ.startswith("Attn"):
                    blk.attn.append(_AttnPP(self, f"down_blocks.{i}.attentions.{j}", ch, hd, 32))
            if i != len(boc) - 1:
                blk.down = _ResnetPP(self, f"down_blocks.{i}.resnet_down", ch, ch, _g(ch), down=True, use_in_shortcut=True)
                blk.skip_conv = f"down_blocks.{i}.skip_conv"
                self._decl(blk.skip_conv + ".weight", (ch, in_channels, 1, 1), fan_in=in_channels)
                self._decl(blk.skip_conv + ".bias", (ch,), fan_in=in_channels, is_bias=True)
                blk.ch = ch
            self.down.append(blk)
        self.mid = SimpleNamespace(
            r0=_ResnetPP(self, "mid_block.resnets.0", ch, ch, _g(ch), scale=mid_block_scale_factor),
            attn=_AttnPP(self, "mid_block.attentions.0", ch, hd, _g(ch), rescale=mid_block_scale_factor),
            r1=_ResnetPP(self, "mid_block.resnets.1", ch, ch, _g(ch), scale=mid_block_scale_factor))
        rev = boc[::-1]
        self.up = []
        out_ch = rev[0]
        for i, typ in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            blk = SimpleNamespace(res=[], attn=None, up=None, skip_norm=None, skip_conv=None, h_ch=[], s_ch=[])
            for j in range(layers_per_block + 1):
                s_ch = in_ch if j == layers_per_block else out_ch
                h_ch = prev if j == 0 else out_ch
                blk.res.append(_ResnetPP(self, f"up_blocks.{i}.resnets.{j}", h_ch + s_ch, out_ch, _g(h_ch + s_ch), _g(out_ch)))
                blk.h_ch.append(h_ch)
                blk.s_ch.append(s_ch)
            if typ.startswith("Attn"):
                blk.attn = _AttnPP(self, f"up_blocks.{i}.attentions.0", out_ch, hd, 32)
            if i != len(boc) - 1:
                blk.up = _ResnetPP(self, f"up_blocks.{i}.resnet_up", out_ch, out_ch, _g(out_ch), _g(out_ch), up=True, use_in_shortcut=True)
                blk.skip_conv = _Conv(self, f"up_blocks.{i}.skip_conv", out_ch, out_channels)
                blk.skip_norm = _Norm(self, f"up_blocks.{i}.skip_norm", out_ch, True, _g(out_ch))
            self.up.append(blk)
        self.norm_out = _Norm(self, "conv_norm_out", boc[0], True, _g(boc[0]))
        self._conv_out = _Conv(self, "conv_out", boc[0], out_channels)
        self._materialise()
        self.time_proj.weight.requires_grad_(False)          # GaussianFourierProjection.weight is a fixed random feature

    # ---- parameter plumbing differences ----
    def _decl(self, name, shape, fan_in=None, is_bias=False, ones=False, zeros=False, fourier=False):
        self._decls.append((name, tuple(shape), dict(fan_in=fan_in, is_bias=is_bias, ones=ones, zeros=zeros, fourier=fourier)))

    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None):
        super().reset_parameters(seed)
        gen = torch.Generator().manual_seed((seed or 0) + 7919) if seed is not None else None
        off, n, _ = self._offs["time_proj.weight"]
        self.flat_param[off:off + n].copy_(torch.randn(n, generator=gen) * 16.0)     # GaussianFourierProjection(scale=16)

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if k != "time_proj.W"}            # alias of time_proj.weight upstream
        return super().load_state_dict(sd, strict)

    # ---- forward / backward launch sequences ----
    def _cat(self, h, s):
        B, ch, H, W = h.shape
        buf = torch.empty((B, ch + s.shape[1], H, W), device=h.device, dtype=torch.float32)
        ops.add_strided(buf[:, :ch], h, accumulate=False)
        ops.add_strided(buf[:, ch:], s, accumulate=False)
        return buf

    def _run_forward(self, x, t, save):
        dev = self._dev
        B, _, S, _ = x.shape
        st = SimpleNamespace(B=B, tape=[], marks={})
        tp = st.tape
        # time embedding: Fourier features of log(sigma) -> MLP -> all time_emb_proj in one GEMM
        four = torch.empty((B, self.time_dim0), device=dev, dtype=torch.float32)
        ops.fourier_embedding(t, self.P["time_proj.weight"], four)
        e1 = torch.empty((B, self.temb_dim), device=dev, dtype=torch.float32)
        ops.linear(four, self.P["time_embedding.linear_1.weight"], self.P["time_embedding.linear_1.bias"], e1)
        e1a = ops.silu_fwd(e1, torch.empty_like(e1))
        emb = torch.empty_like(e1)
        ops.linear(e1a, self.P["time_embedding.linear_2.weight"], self.P["time_embedding.linear_2.bias"], emb)
        emb_act = ops.silu_fwd(emb, torch.empty_like(emb))
        st.temb_all = torch.empty((B, self.temb_total), device=dev, dtype=torch.float32)
        ops.linear(emb_act, self.Wt_all, self.bt_all, st.temb_all)
        if save:
            st.temb_saved = (four, e1, e1a, emb, emb_act)
        # ---- down ----
        skip = x
        h = torch.empty((B, self._conv_in.cout, S, S), device=dev, dtype=torch.float32)
        self._conv_in.fwd(x, h)
        tp.append(("conv_in", x))
        res = [h]
        for blk in self.down:
            for j, r in enumerate(blk.res):
                h, s = r.fwd(h, st, save)
                tp.append(("res", r, s))
                if blk.attn:
                    h, s = blk.attn[j].fwd(h, st, save)
                    tp.append(("attn", blk.attn[j], s))
                res.append(h)
                tp.append(("skip_out",))
            if blk.down is not None:
                h2, s = blk.down.fwd(h, st, save)
                tp.append(("res", blk.down, s))
                sk2 = torch.empty((B, skip.shape[1], skip.shape[2] // 2, skip.shape[3] // 2), device=dev, dtype=torch.float32)
                ops.fir_resample2(skip.contiguous(), sk2, up=False)
                skip = sk2
                h = torch.empty_like(h2)
                ops.conv1x1(skip, self.P[blk.skip_conv + ".weight"].view(blk.ch, -1), self.P[blk.skip_conv + ".bias"], h, residual=h2)
                tp.append(("skip_conv_down", blk, skip))
                res.append(h)
                tp.append(("skip_out",))
        st.marks["down_end"] = len(tp)
        # ---- mid ----
        h, s = self.mid.r0.fwd(h, st, save); tp.append(("res", self.mid.r0, s))
        h, s = self.mid.attn.fwd(h, st, save); tp.append(("attn", self.mid.attn, s))
        h, s = self.mid.r1.fwd(h, st, save); tp.append(("res", self.mid.r1, s))
        st.marks["mid_end"] = len(tp)
        # ---- up ----
        sk = None
        for blk in self.up:
            for j, r in enumerate(blk.res):
                s_t = res.pop()
                cat = self._cat(h, s_t)
                tp.append(("cat", h.shape[1]))
                h, s = r.fwd(cat, st, save)
                tp.append(("res", r, s))
            if blk.attn is not None:
                h, s = blk.attn.fwd(h, st, save)
                tp.append(("attn", blk.attn, s))
            if sk is not None:
                up = torch.empty((B, sk.shape[1], 2 * sk.shape[2], 2 * sk.shape[3]), device=dev, dtype=torch.float32)
                ops.fir_resample2(sk, up, up=True)
                sk = up
                tp.append(("skip_up",))
            if blk.up is not None:
                a = torch.empty_like(h)
                mo, ro = blk.skip_norm.fwd(h, a)
                ns = torch.empty((B, self.out_channels, h.shape[2], h.shape[3]), device=dev, dtype=torch.float32)
                blk.skip_conv.fwd(a, ns, residual=sk)
                sk = ns
                tp.append(("skip_conv_up", blk, h, a, mo, ro))
                h, s = blk.up.fwd(h, st, save)
                tp.append(("res", blk.up, s))
        assert not res
        a = torch.empty_like(h)
        mo, ro = self.norm_out.fwd(h, a)
        y = torch.empty((B, self.out_channels, S, S), device=dev, dtype=torch.float32)
        self._conv_out.fwd(a, y)
        if sk is not None:
            ops.add_strided(y, sk, accumulate=True)
        out = torch.empty_like(y)
        ops.rowscale(y, t, out, divide=True)                     # UNet2DModel: sample / timesteps (sigma)
        if save:
            st.out_saved = (h, a, mo, ro, t, sk is not None)
        else:
            st.tape = None
        return out, st

    def _run_backward(self, st, dout):
        dev = self._dev
        B = st.B
        self._prepare_backward(B)
        self._cs_begin(B)
        st.d_temb_all = self._d_temb_buffer(B)
        hook = self.bucket_ready_hook
        h_in, a, mo, ro, t, has_sk = st.out_saved
        dy = torch.empty_like(dout)
        ops.rowscale(dout.contiguous(), t, dy, divide=True)
        dsk = dy if has_sk else None                           # gradient wrt the output-image pyramid (skip_sample)
        da = torch.empty_like(a)
        self._conv_out.bwd(dy, a, da)
        g = torch.empty_like(h_in)
        self.norm_out.bwd(da, h_in, mo, ro, g)
        tp = st.tape
        skip_grads: List[torch.Tensor] = []                   # gradients of the down-path skip tensors, first produced first
        while tp:
            if hook is not None:
                if len(tp) == st.marks["mid_end"]:
                    self._cs_flush()
                    hook(0)
                elif len(tp) == st.marks["down_end"]:
                    self._cs_flush()
                    hook(1)
            rec = tp.pop()
            kind = rec[0]
            if kind == "res":
                g = rec[1].bwd(rec[2], g, st)
            elif kind == "attn":
                g = rec[1].bwd(rec[2], g, st)
            elif kind == "skip_conv_up":
                _, blk, h, a2, m2, r2 = rec
                # skip_sample_new = skip_conv(silu(gn(h))) + skip_sample_up ; dsk flows to both
                da2 = torch.empty_like(a2)
                blk.skip_conv.bwd(dsk, a2, da2)
                gh = torch.empty_like(h)
                blk.skip_norm.bwd(da2, h, m2, r2, gh, extra=g)       # g: gradient from resnet_up's input (same h)
                g = gh
            elif kind == "skip_up":
                nd = torch.empty((B, dsk.shape[1], dsk.shape[2] // 2, dsk.shape[3] // 2), device=dev, dtype=torch.float32)
                ops.fir_resample2(dsk.contiguous(), nd, up=False, scale=4.0)
                dsk = nd
            elif kind == "cat":
                ch = rec[1]
                skip_grads.append(g[:, ch:])
                gg = torch.empty((B, ch, g.shape[2], g.shape[3]), device=dev, dtype=torch.float32)
                ops.add_strided(gg, g[:, :ch], accumulate=False)
                g = gg
            elif kind == "skip_out":
                # this tensor was also consumed by the up path (last produced = first consumed = appended last): add that gradient
                ops.add_strided(g, skip_grads.pop(), accumulate=True)
            elif kind == "skip_conv_down":
                _, blk, skimg = rec
                # h = skip_conv(skimg) + h2: weight / bias gradients of the 1x1 conv; the image pyramid has no parameters upstream
                ops.conv_wgrad(g, skimg, self.G[blk.skip_conv + ".weight"].view(blk.ch, -1), B_PLAIN, self.wgrad_ws, accumulate=True)
                ws = self.scratch_bc(B, blk.ch)
                ops.rowsum(g, ws)
                self.colsum_later(ws, self.G[blk.skip_conv + ".bias"], B, blk.ch)
            elif kind == "conv_in":
                ops.add_strided(g, skip_grads.pop(), accumulate=True)
                self._conv_in.bwd(g, rec[1], None)
            else:
                raise RuntimeError(kind)
        assert not skip_grads
        if hook is not None:
            self._cs_flush()
            hook(2)
        four, e1, e1a, emb, emb_act = st.temb_saved
        d = st.d_temb_all
        ops.linear_wgrad(d, emb_act, self.gWt_all, accumulate=True)
        self.colsum_later(d, self.gbt_all, B, self.temb_total)
        d_act = torch.empty_like(emb_act)
        ops.linear_dgrad(d, self.Wt_all, d_act)
        d_emb = ops.silu_bwd(d_act, emb, torch.empty_like(emb))
        ops.linear_wgrad(d_emb, e1a, self.G["time_embedding.linear_2.weight"], accumulate=True)
        ops.colsum(d_emb, self.G["time_embedding.linear_2.bias"], B, self.temb_dim, accumulate=True)
        d_e1a = torch.empty_like(e1a)
        ops.linear_dgrad(d_emb, self.P["time_embedding.linear_2.weight"], d_e1a)
        d_e1 = ops.silu_bwd(d_e1a, e1, torch.empty_like(e1))
        ops.linear_wgrad(d_e1, four, self.G["time_embedding.linear_1.weight"], accumulate=True)
        ops.colsum(d_e1, self.G["time_embedding.linear_1.bias"], B, self.temb_dim, accumulate=True)
        self._cs_flush()
        if hook is not None:
            hook(3)
