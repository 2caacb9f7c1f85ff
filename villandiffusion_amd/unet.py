"""MI355X-native ``UNet2DModel`` (M1 in SURVEY.md §8a): same constructor arguments, ``state_dict`` names,
``forward(x, t, return_dict=False) -> (eps_hat,)`` contract and attributes (``in_channels``, ``sample_size``,
``config``) as the diffusers class the reference instantiates (model.py:603-605, 816-834) -- but every op is one of
the hand-written gfx950 kernels behind the C ABI, and forward AND backward are explicit launch sequences
(no autograd graph inside the network):

* all parameters are views into ONE flat fp32 buffer (and their ``.grad`` into one flat gradient buffer), so the
  optimiser and the RCCL all-reduce work on a single contiguous bucket; the 22 ``time_emb_proj`` matrices are laid out
  adjacently and run as ONE GEMM, and to_q/to_k/to_v of each attention block are one [3C, C] matrix;
* skip connections are zero-copy: the producer of a skip tensor writes straight into the channel slice of the
  concat buffer its up-block consumer will read (kernels take a batch stride);
* bias, timestep-embedding broadcast and residual adds live in the GEMM epilogues; GroupNorm+SiLU is one kernel;
  nearest-2x upsampling and the asymmetric stride-2 padding are folded into the convolution's gather.

Backward writes parameter gradients (accumulating) into the flat gradient buffer as a side effect of
``loss.backward()``; call ``zero_grad()`` (one kernel) between optimiser steps.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .lib import B_CONV3, B_CONV3_DIL, B_CONV3_S2, B_CONV3_T, B_CONV3_UP, B_PLAIN, A_COL, A_ROW, B_KCONTIG

LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class _Node(nn.Module):
    """Anonymous container used to reproduce diffusers' dotted state-dict names."""


def _ensure_path(root: nn.Module, parts: Sequence[str]) -> nn.Module:
    m = root
    for p in parts:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    return m


# ----------------------------------------------------------------------------------------------------------- layers
CONV_MATH_DEFAULT = os.environ.get("VILLAN_CONV_MATH", "bf16x3")
if CONV_MATH_DEFAULT not in ("bf16x3", "f32", "f16"):
    raise ValueError(f"VILLAN_CONV_MATH must be 'bf16x3', 'f32' or 'f16', got {CONV_MATH_DEFAULT!r}")


class _PackedConvWeights:
    """Operands of the split-precision ("bf16x3") 3x3 convolution kernel (vd_gemm_desc.a_packed): every eligible convolution's
    weights as bf16 (hi, lo) pairs in MFMA fragment order -- one image for the forward convolution and one (transposed) for the
    stride-1 input gradient.  Each set is rebuilt in ONE launch when the weights have changed since it was built."""

    def __init__(self, net, f16: bool = False):
        self.net = net
        self.f16 = f16                                   # round 4, opt-in mixed precision: ONE f16 plane per operand (half the bytes), vd_gemm_desc.math = 2
        self.off = {False: {}, True: {}}                 # bwd? -> key -> (offset, n) in int32 elements
        self.jobs = {False: [], True: []}                # (key, source tensor getter, M, C, taps, row_stride, chan_stride)
        self.total = 0
        cands = []                                       # (key, getter of the [cout, cin*taps] weights, cout, cin, taps)
        for name, shape, _ in net._layout:
            if name.endswith(".weight") and len(shape) == 4 and shape[2] == shape[3] and shape[2] in (1, 3):
                cands.append((name[:-7], (lambda n=name: net.P[n]), shape[0], shape[1], shape[2] * shape[3]))
        for prefix, ch in getattr(net, "_qkv", []):      # attention projections: fused q/k/v [3C, C] and to_out [C, C] (1x1 convolutions)
            if prefix + "::qkv_w" in getattr(net, "Pq", {}):
                cands.append((prefix + "::qkv", (lambda k=prefix + "::qkv_w": net.Pq[k]), 3 * ch, ch, 1))
            if prefix + ".to_out.0.weight" in net.P:
                cands.append((prefix + ".to_out.0", (lambda n=prefix + ".to_out.0.weight": net.P[n]), ch, ch, 1))
        plan = []
        for key, get, cout, cin, T in cands:
            for bwd, (M, Cc, rs, cs) in ((False, (cout, cin, cin * T, T)), (True, (cin, cout, T, cin * T))):
                plan.append((key, get, bwd, M, Cc, T, rs, cs))
            if T == 9 and ".downsamplers." in key:
                # stride-2 input gradient = plain product G[c*9+t][pixel] = sum_m W[m][c*9+t] dY[m][pixel] followed by col2im (ops.conv3x3_s2_dgrad):
                # its A operand is the [cin*9, cout] transpose of the weight matrix, packed as a one-tap operand
                plan.append((key + "::s2g", get, True, cin * 9, cout, 1, 1, cin * 9))
        for key, get, bwd, M, Cc, T, rs, cs in plan:
            if Cc % 16 == 0 and M >= 64:
                n = (M + 127) // 128 * 128 * Cc * T // (2 if f16 else 1)
                self.off[bwd][key] = (self.total, n)
                self.jobs[bwd].append((key, get, M, Cc, T, rs, cs))
                self.total += n
        self.buf: Optional[torch.Tensor] = None
        self.tables = {}
        self.key = {False: None, True: None}

    def view(self, prefix, bwd):
        """Packed operand of `prefix` (None if that convolution is not eligible), fresh for the current weights."""
        ent = self.off[bwd].get(prefix)
        if ent is None:
            return None
        self.refresh(bwd)
        return self.buf[ent[0]:ent[0] + ent[1]]

    def refresh(self, bwd):
        """Rebuild the `bwd` set (one launch) if the weights have changed since it was built."""
        net = self.net
        if not self.jobs[bwd]:
            return
        key = (net.flat_param._version, ops.WEIGHTS_EPOCH)
        if self.key[bwd] != key:
            if self.buf is None:
                self.buf = torch.empty(self.total, device=net.flat_param.device, dtype=torch.int32)
            if bwd not in self.tables:
                rows, blk = [], 0
                for k, get, M, Cc, T, rs, cs in self.jobs[bwd]:
                    o, n = self.off[bwd][k]
                    rows.append([get().data_ptr(), self.buf.data_ptr() + 4 * o, M, Cc, rs, cs, blk, T])
                    blk += ((M + 127) // 128 * 128 * (Cc // 16) * 2 + 255) // 256
                self.tables[bwd] = (torch.tensor(rows, dtype=torch.int64).to(self.buf.device), len(rows), blk)
            tab, nj, blk = self.tables[bwd]
            (ops.conv3_pack_weights_f16_multi if self.f16 else ops.conv3_pack_weights_multi)(tab, nj, blk)
            self.key[bwd] = key


def _split(net) -> bool:
    """Contractions on the bf16 matrix cores: "bf16x3" (default), and "f16" -- the opt-in mixed-precision mode (round 4), which ADDITIONALLY runs
    the full-size 3x3 / 1x1 forward and input-gradient contractions as single f16 products (everything else as in "bf16x3")."""
    return getattr(net, "conv_math", "f32") in ("bf16x3", "f16")


def _with_f16(net, pk, key, bwd):
    """In "f16" mode the operand handed to ops.gemm is the pair (split-precision operand, f16 operand); ops.gemm picks per problem."""
    if pk is None or getattr(net, "conv_math", "f32") != "f16":
        return pk
    p16 = getattr(net, "_packed16", None)
    if p16 is None:
        p16 = net._packed16 = _PackedConvWeights(net, f16=True)
    v16 = p16.view(key, bwd)
    return pk if v16 is None else (pk, v16)


def _bx3_packed(net, prefix, bwd, M, Cc, OH, OW, mode):
    """The packed operand to hand to ops.conv3x3 for this call, or None (exact-f32 kernels)."""
    if not _split(net) or not ops.bx3_eligible(M, Cc, OH, OW, mode):
        return None
    pk = getattr(net, "_packed", None)
    if pk is None:
        pk = net._packed = _PackedConvWeights(net)
    return _with_f16(net, pk.view(prefix, bwd), prefix, bwd)


def _bx3_packed_1x1(net, key, bwd, M, K, NP, nb=None):
    """Packed operand of a 1x1 convolution / projection for ops.conv1x1 / ops.gemm(a_packed=...), or None (exact-f32 kernels)."""
    if not _split(net) or not ops.gemm_bx3_eligible(M, K, NP, nb):
        return None
    pk = getattr(net, "_packed", None)
    if pk is None:
        pk = net._packed = _PackedConvWeights(net)
    return _with_f16(net, pk.view(key, bwd), key, bwd)


def _amath(net, M, K, NP) -> int:
    """vd_gemm_desc.math for a product of two activation matrices (attention scores / values and their gradients)."""
    return int(_split(net) and ops.gemm_bx3_act_eligible(M, K, NP))


def _wgrad1x1_math(net, dy, x) -> int:
    """vd_wgrad_desc.math for a 1x1 weight gradient dW[M, C] = sum dy x^T."""
    ok = _split(net) and ops.wgrad_bx3_eligible(dy.shape[1], x.shape[1], dy.shape[2], dy.shape[3], B_PLAIN) \
        and x.stride(0) % 4 == 0 and dy.stride(0) % 4 == 0
    return int(ok)


class _Conv:
    """3x3 convolution (mode selects the gather) with bias; weight stored [M, C, 3, 3] like diffusers."""

    def __init__(self, net, prefix, cin, cout, mode=B_CONV3, pad=0):
        self.net, self.prefix, self.cin, self.cout, self.mode, self.pad = net, prefix, cin, cout, mode, pad
        net._decl(prefix + ".weight", (cout, cin, 3, 3), fan_in=cin * 9)
        net._decl(prefix + ".bias", (cout,), fan_in=cin * 9, is_bias=True)

    def w2d(self):
        return self.net.P[self.prefix + ".weight"].view(self.cout, self.cin * 9)

    def fwd(self, x, out, rowadd=None, rowadd_bstride=0, residual=None, gn_ss=None, gn_part=None, act_out=None):
        """gn_part: buffer for the per-tile channel sums of `out` (ops.conv3x3); ops.GN_PART_WRITTEN says whether this launch filled it."""
        pk = _bx3_packed(self.net, self.prefix, False, self.cout, self.cin, out.shape[2], out.shape[3], self.mode)
        if gn_ss is not None and (out.shape[3] not in (16, 32) or self.mode != B_CONV3):
            pk = None                                      # the folded-GroupNorm loader of the split-precision kernel: 16x16 / 32x32 only
        return ops.conv3x3(x, self.w2d(), self.net.P[self.prefix + ".bias"], out, mode=self.mode, rowadd=rowadd,
                           rowadd_bstride=rowadd_bstride, residual=residual, pad=self.pad, gn_ss=gn_ss, a_packed=pk, gn_part=gn_part, act_out=act_out)

    def us_input(self, h, save):
        """Upsample2D's convolution (round 6): its input -- a block output, f32 -- as a pre-split image when the persistent kernel reads one for this shape.
        The pack pass (one read, one write of the HALF-resolution tensor) used to run on the weight-gradient stream for the weight gradient alone; made
        here, the forward convolution copies (hi, lo) units too instead of splitting every element once per tap row and channel tile."""
        net = self.net
        B, _, H, W = h.shape
        if (self.mode != B_CONV3_UP or not net.us_fwd_presplit or not getattr(net, "presplit", False) or net.conv_math != "bf16x3" or H != W
                or not ops.conv_presplit_ok(B, self.cin, self.cout, 2 * H, 2 * W, B_CONV3_UP)
                or (save and not (net.group_wgrad and ops.wgrad_presplit_ok(B, self.cin, self.cout, 2 * H, B_CONV3_UP)))):
            return h
        return ops.presplit_pack(h)

    def bwd(self, dout, x, dx, bias_ws=None, skip_bias=False, dout_ps=None):
        """dW, db (accumulated into the flat gradient) and, if dx is given, the input gradient.  x a PreSplit image (round 5): the weight gradient
        then wants dY pre-split too -- `dout_ps` when the producer of dout wrote one (dout itself may then be None), else a pack pass queued on the
        weight-gradient stream -- and the input gradient reads `dout_ps` when it is there."""
        net = self.net
        dout_f32 = dout                                       # (the bias gradient's row sums read the f32 tensor)
        if isinstance(x, ops.PreSplit):
            dy_ps = dout_ps if dout_ps is not None else net.pack_later(dout)
            net.wgrad(dy_ps, x, net.G[self.prefix + ".weight"].view(self.cout, self.cin * 9), self.mode, pad=self.pad, math_mode=1)
            if dout_ps is not None:
                dout = dout_ps                                # the input gradient copies (hi, lo) units too
        elif (self.mode == B_CONV3_UP and getattr(net, "presplit", False) and net.conv_math == "bf16x3" and net.group_wgrad
              and dout.shape[2] == dout.shape[3] and ops.wgrad_presplit_ok(dout.shape[0], self.cin, self.cout, dout.shape[3], B_CONV3_UP)):
            # Upsample2D's convolution: its input is a block output (f32), so the image is packed on the weight-gradient stream -- two small passes
            # off the critical path for the LDS-DMA kernel (the 256 -> 256 layer at 32x32 outputs is 0.42 ms per step on the converting kernel)
            dy_ps = dout_ps if dout_ps is not None else net.pack_later(dout)
            net.wgrad(dy_ps, net.pack_later(x), net.G[self.prefix + ".weight"].view(self.cout, self.cin * 9), self.mode, pad=self.pad, math_mode=1)
        else:
            assert dout is not None
            bx3 = _split(net) and ops.wgrad_bx3_eligible(self.cout, self.cin, dout.shape[2], dout.shape[3],
                                                                                           self.mode) and x.stride(0) % 4 == 0
            net.wgrad(dout, x, net.G[self.prefix + ".weight"].view(self.cout, self.cin * 9), self.mode, pad=self.pad, math_mode=int(bx3))
        if not skip_bias:
            B = dout.shape[0]
            ws = bias_ws if bias_ws is not None else net.scratch_bc(B, self.cout)
            if bias_ws is None:
                assert dout_f32 is not None
                net.rowsum(dout_f32, ws)
            net.colsum_later(ws, net.G[self.prefix + ".bias"], B, self.cout, ld=(ws.stride(0) if ws.dim() == 2 else self.cout))
        if dx is None:
            return None
        if self.mode == B_CONV3_S2:
            pk = _bx3_packed_1x1(net, self.prefix + "::s2g", True, self.cin * 9, self.cout, dout.shape[2] * dout.shape[3], dout.shape[0])
            return ops.conv3x3_s2_dgrad(dout, self.w2d(), dx, pad=self.pad, a_packed=pk)
        # split-precision dgrad reads the packed transposed operand; the f32 transposed copy is then only a shape carrier
        pk = _bx3_packed(net, self.prefix, True, self.cin, self.cout, dout.shape[2], dout.shape[3], B_CONV3_T) \
            if self.mode in (B_CONV3, B_CONV3_UP) else None
        wt = net.wt_view(self.prefix, self.cout, self.cin, 9, fresh=pk is None)     # [C, M*9]
        if self.mode == B_CONV3:
            ops.conv3x3(dout, wt, None, dx, mode=B_CONV3_T, a_packed=pk)
        elif self.mode == B_CONV3_UP:
            B, _, OH, OW = dout.shape
            if pk is not None and ops.bx3_pool2_eligible(self.cin, self.cout, OH, OW, B):
                # (round 6: the pre-split image of dout -- written for the weight gradient by dout's producer -- feeds the input gradient too)
                src = dout_ps if (dout_ps is not None and net.us_dgrad_presplit and ops.conv_presplit_ok(B, self.cout, self.cin, OH, OW, B_CONV3_T)) else dout
                ops.conv3x3(src, wt, None, dx, mode=B_CONV3_T, a_packed=pk, pool2=True)     # 2x2 sums in the epilogue: no 4x tensor
            else:
                dU = torch.empty((B, self.cin, OH, OW), device=dout.device, dtype=torch.float32)
                ops.conv3x3(dout, wt, None, dU, mode=B_CONV3_T, a_packed=pk)
                ops.sumpool2x2(dU, dx)
        else:
            raise NotImplementedError(self.mode)
        return dx


class _Norm:
    def __init__(self, net, prefix, ch, silu, groups=None):
        self.net, self.prefix, self.ch, self.silu = net, prefix, ch, silu
        self.groups = net.groups if groups is None else groups       # NCSN++ sizes the groups per layer: min(ch // 4, 32)
        net._decl(prefix + ".weight", (ch,), ones=True)
        net._decl(prefix + ".bias", (ch,), zeros=True)

    def fwd(self, x, y):
        net = self.net
        B = x.shape[0]
        mean = torch.empty(B * self.groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops.groupnorm_fwd(x, net.P[self.prefix + ".weight"], net.P[self.prefix + ".bias"], y, mean, rstd, self.groups,
                          net.eps, self.silu)
        return mean, rstd

    def fwd_ps(self, x):
        """y = silu?(gn(x)) as a PRE-SPLIT image (ops.PreSplit): what the 3x3 convolution and its weight gradient read without converting."""
        net = self.net
        B = x.shape[0]
        mean = torch.empty(B * self.groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        y = ops.presplit_empty(x.shape, x.device)
        ops.groupnorm_fwd_presplit(x, net.P[self.prefix + ".weight"], net.P[self.prefix + ".bias"], y, mean, rstd, self.groups, net.eps, self.silu)
        return y, mean, rstd

    def ps_ok(self, HW: int) -> bool:
        return ops.groupnorm_presplit_ok(self.ch, HW, self.groups)

    def stats(self, x, full=False):
        """Statistics-only pass: [B, C, 2] scale / shift pairs consumed by _Conv.fwd(gn_ss=...) (inference path; round 4: the training forward
        too).  full: also the (mean, rstd) the backward pass needs."""
        net = self.net
        B = x.shape[0]
        ss = torch.empty((B, self.ch, 2), device=x.device, dtype=torch.float32)
        mean = torch.empty(B * self.groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops.groupnorm_stats(x, net.P[self.prefix + ".weight"], net.P[self.prefix + ".bias"], ss, mean, rstd, self.groups, net.eps)
        return (ss, mean, rstd) if full else ss

    def fwd_later(self, x):
        """silu(gn(x)) for a weight gradient of the backward pass, computed on the weight-gradient side stream right before the grouped launch
        that reads it (UNet2DModel.gn_later): the training forward folds the normalisation into the convolution's loader, so this tensor is
        needed by nobody else and never sits on the critical path."""
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        self.net.gn_later(self, x, y)
        return y

    def _fwd_now(self, x, y):
        B = x.shape[0]
        mean = torch.empty(B * self.groups, device=x.device, dtype=torch.float32)
        ops.groupnorm_fwd(x, self.net.P[self.prefix + ".weight"], self.net.P[self.prefix + ".bias"], y, mean, torch.empty_like(mean),
                          self.groups, self.net.eps, self.silu)

    def stats_from_partials(self, part, tiles, B, HW, full=False):
        """stats() of a tensor whose producing convolution left its per-tile channel sums in `part` (vd_gemm_desc.gn_part)."""
        net = self.net
        ss = torch.empty((B, self.ch, 2), device=part.device, dtype=torch.float32)
        mean = torch.empty(B * self.groups, device=part.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops.groupnorm_stats_from_partials(part, tiles, net.P[self.prefix + ".weight"], net.P[self.prefix + ".bias"], ss, mean, rstd, HW,
                                          self.groups, net.eps)
        return (ss, mean, rstd) if full else ss

    def bwd(self, dy, x, mean, rstd, dx, extra=None, extra2=None, rowsum=None, dx_ps=None):
        """extra / extra2: residual gradients added into dx; rowsum ([B, C] view, row stride free): per-image channel sums of the dx
        written, i.e. the bias-gradient rows of the layer that produced x, from the same pass.  dx_ps (ops.PreSplit): dx ALSO (dx given) or ONLY
        (dx None) as a pre-split image -- the operand of the producing convolution's input and weight gradients (needs ps_ok(HW))."""
        net = self.net
        B = x.shape[0]
        wg, wb = net.scratch_bc(B, self.ch, 1), net.scratch_bc(B, self.ch, 2)
        if dx_ps is not None:
            ops.groupnorm_bwd_presplit(dy, x, mean, rstd, net.P[self.prefix + ".weight"], net.P[self.prefix + ".bias"], dx, dx_ps, wg, wb,
                                       self.groups, self.silu, extra=extra, extra2=extra2, rowsum=rowsum,
                                       rowsum_ld=(rowsum.stride(0) if rowsum is not None else None))
        else:
            ops.groupnorm_bwd(dy, x, mean, rstd, net.P[self.prefix + ".weight"], net.P[self.prefix + ".bias"], dx, wg, wb,
                              self.groups, self.silu, extra=extra, extra2=extra2, rowsum=rowsum,
                              rowsum_ld=(rowsum.stride(0) if rowsum is not None else None))
        net.colsum_later(wg, net.G[self.prefix + ".weight"], B, self.ch)
        net.colsum_later(wb, net.G[self.prefix + ".bias"], B, self.ch)
        return dx


class _Resnet:
    """ResnetBlock2D: conv1(silu(gn(x))) + temb -> conv2(silu(gn(.))) + shortcut(x)."""

    def __init__(self, net, prefix, cin, cout):
        self.net, self.prefix, self.cin, self.cout = net, prefix, cin, cout
        self.norm1 = _Norm(net, prefix + ".norm1", cin, True)
        self.conv1 = _Conv(net, prefix + ".conv1", cin, cout)
        self.temb_off = net._decl_temb(prefix + ".time_emb_proj", cout)
        self.norm2 = _Norm(net, prefix + ".norm2", cout, True)
        self.conv2 = _Conv(net, prefix + ".conv2", cout, cout)
        self.has_sc = cin != cout
        self._ps_cache = {}
        if self.has_sc:
            net._decl(prefix + ".conv_shortcut.weight", (cout, cin, 1, 1), fan_in=cin)
            net._decl(prefix + ".conv_shortcut.bias", (cout,), fan_in=cin, is_bias=True)

    def _sc_dgrad(self, dout, B, H, W, dev):
        """dsc = W_sc^T dout: the shortcut's input gradient."""
        net = self.net
        wsc = net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin)
        dsc = torch.empty((B, self.cin, H, W), device=dev, dtype=torch.float32)
        HW = H * W
        ops.gemm(wsc, dout, dsc, M=self.cin, N=B * HW, K=self.cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=self.cin,
                 ldb=HW, b_bstride=ops._img(dout)[4], ldd=HW, d_bstride=self.cin * HW,
                 a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", True, self.cin, self.cout, HW, B))
        return dsc

    def ps_plan(self, B, H, W) -> bool:
        """Pre-split operands for this block's two 3x3 convolutions in a TRAINING pass (round 5): both GroupNorms have a pre-split producer kernel,
        all four convolution launches (conv1 / conv2 forward and input gradient) go to the persistent 16x16x32 kernel -- the only reader of
        pre-split images -- and both weight gradients have a pre-split grouped kernel.  Decided per (block, batch, image size), cached."""
        net = self.net
        if not getattr(net, "presplit", False) or net.conv_math != "bf16x3" or not net.group_wgrad or H != W:
            return False
        key = (B, H, W)
        ok = self._ps_cache.get(key)
        if ok is None:
            ci, co = self.cin, self.cout
            ok = (self.norm1.ps_ok(H * W) and self.norm2.ps_ok(H * W)
                  and ops.conv_presplit_ok(B, ci, co, H, W, B_CONV3) and ops.conv_presplit_ok(B, co, co, H, W, B_CONV3)
                  and ops.conv_presplit_ok(B, co, ci, H, W, B_CONV3_T) and ops.conv_presplit_ok(B, co, co, H, W, B_CONV3_T)
                  and ops.wgrad_presplit_ok(B, ci, co, H) and ops.wgrad_presplit_ok(B, co, co, H))
            self._ps_cache[key] = ok
        return ok

    def fwd(self, x, out, st, save):
        net = self.net
        B, _, H, W = x.shape
        dev = x.device
        fuse = net.fuse_gn_inference if net.fuse_gn_inference is not None else _split(net)
        # no-grad forward: the folded loader normalises + splits every activation once per 128-channel tile of the OUTPUT; from two tiles on, the
        # pre-split producer (one pass, one split) + the copying loader are cheaper (profiles/r06_sampler_presplit_ab.txt: +0.9 % sampler throughput
        # with the 16x16 level's 256-channel blocks on this path; the 128-channel 32x32 blocks lose 1.4 % on it and stay folded)
        ps_ng = not save and net.nograd_presplit and self.cout > 128 and self.ps_plan(B, H, W)
        if not save and not ps_ng and fuse and ops.gn_fusable(x, self.cout) and self.cin * H * W // net.groups <= 12288 \
                and self.cout * H * W // net.groups <= 12288:
            # inference: GroupNorm + SiLU folded into the convolutions' patch loaders -- a statistics pass (one read) replaces the
            # normalise pass (read + write) and the normalised activations never reach HBM
            h1 = torch.empty((B, self.cout, H, W), device=dev, dtype=torch.float32)
            # norm2's statistics come out of conv1's epilogue (per-tile channel sums) when conv1 runs on the 16x16x32 kernel: no read of h1
            tiles = (H * W) // 256
            part = torch.empty((B, tiles, self.cout, 2), device=dev, dtype=torch.float32) if (getattr(net, "gn_stats_in_epilogue", False) and tiles > 0) else None
            self.conv1.fwd(x, h1, rowadd=st.temb_all[:, self.temb_off:], rowadd_bstride=st.temb_all.stride(0), gn_ss=self.norm1.stats(x),
                           gn_part=part)
            if part is not None and ops.GN_PART_WRITTEN:
                ss2 = self.norm2.stats_from_partials(part, tiles, B, H * W)
            else:
                ss2 = self.norm2.stats(h1)
            if self.has_sc:
                ops.conv1x1(x, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                            net.P[self.prefix + ".conv_shortcut.bias"], out,
                            a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", False, self.cout, self.cin, H * W, B))
                self.conv2.fwd(h1, out, residual=out, gn_ss=ss2)
            else:
                self.conv2.fwd(h1, out, residual=x, gn_ss=ss2)
            return None
        if save and net.fold_gn_train and fuse and net.conv_math == "bf16x3" and not net.defer_gn_fwd and ops.gn_fusable(x, self.cout) \
                and self.cin % 32 == 0 and self.cout % 32 == 0 and self.cin * H * W // net.groups <= 12288 \
                and self.cout * H * W // net.groups <= 12288:
            # Training forward without a normalise pass (round 4): the persistent convolution's loader applies GroupNorm + SiLU (as in the no-grad
            # path) and WRITES the normalised activation it computes anyway (vd_gemm_desc.act_out) -- the operand the weight gradient needs;
            # norm2's statistics come out of conv1's epilogue (gn_part), norm1's from a statistics pass (one read).  Per resnet: two normalise
            # passes (read + write each) become one read; mean / rstd saved for the backward are the ones the forward used.
            tiles = (H * W) // 256
            ss1, m1, r1 = self.norm1.stats(x, full=True)
            a1 = torch.empty((B, self.cin, H, W), device=dev, dtype=torch.float32)
            h1 = torch.empty((B, self.cout, H, W), device=dev, dtype=torch.float32)
            part = torch.empty((B, tiles, self.cout, 2), device=dev, dtype=torch.float32) if tiles > 0 else None
            self.conv1.fwd(x, h1, rowadd=st.temb_all[:, self.temb_off:], rowadd_bstride=st.temb_all.stride(0), gn_ss=ss1, gn_part=part, act_out=a1)
            wrote_a1, wrote_part = ops.ACT_OUT_WRITTEN, part is not None and ops.GN_PART_WRITTEN
            if not wrote_a1:                                      # (a grid the persistent kernel does not take: small batches)
                m1, r1 = self.norm1.fwd(x, a1)
            if wrote_part:
                ss2, m2, r2 = self.norm2.stats_from_partials(part, tiles, B, H * W, full=True)
            else:
                ss2, m2, r2 = self.norm2.stats(h1, full=True)
            a2 = torch.empty_like(h1)
            if self.has_sc:
                ops.conv1x1(x, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                            net.P[self.prefix + ".conv_shortcut.bias"], out,
                            a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", False, self.cout, self.cin, H * W, B))
                self.conv2.fwd(h1, out, residual=out, gn_ss=ss2, act_out=a2)
            else:
                self.conv2.fwd(h1, out, residual=x, gn_ss=ss2, act_out=a2)
            if not ops.ACT_OUT_WRITTEN:
                m2, r2 = self.norm2.fwd(h1, a2)
            return (x, a1, m1, r1, h1, a2, m2, r2)
        if save and net.defer_gn_fwd and fuse and _split(net) and ops.gn_fusable(x, self.cout) \
                and self.cin % 32 == 0 and self.cout % 32 == 0 and self.cin * H * W // net.groups <= 12288 \
                and self.cout * H * W // net.groups <= 12288:
            # training forward (round 4): GroupNorm + SiLU folded into the convolutions' loaders as in the no-grad path -- a statistics pass (one
            # read) instead of the normalise pass (read + write) on the critical path; bit-identical activations (the folded loader evaluates the
            # same expression on the same scale / shift pairs).  silu(gn(.)) itself is only an operand of the WEIGHT gradients: it is recomputed
            # on their side stream in the backward pass (_Norm.fwd_later), where the HBM-bound pass runs beside MFMA-bound kernels.
            ss1, m1, r1 = self.norm1.stats(x, full=True)
            h1 = torch.empty((B, self.cout, H, W), device=dev, dtype=torch.float32)
            self.conv1.fwd(x, h1, rowadd=st.temb_all[:, self.temb_off:], rowadd_bstride=st.temb_all.stride(0), gn_ss=ss1)
            ss2, m2, r2 = self.norm2.stats(h1, full=True)
            if self.has_sc:
                ops.conv1x1(x, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                            net.P[self.prefix + ".conv_shortcut.bias"], out,
                            a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", False, self.cout, self.cin, H * W, B))
                self.conv2.fwd(h1, out, residual=out, gn_ss=ss2)
            else:
                self.conv2.fwd(h1, out, residual=x, gn_ss=ss2)
            return (x, None, m1, r1, h1, None, m2, r2)
        ps = (save and self.ps_plan(B, H, W)) or ps_ng
        # The shortcut (a 1x1 convolution of x, HBM-bound) does not depend on the norm1 -> conv1 -> norm2 chain: with net.sc_stream it runs on an
        # auxiliary stream beside those kernels and is joined before conv2 adds it as the residual
        sc_aux = self.has_sc and save and net.aux_fork()
        if sc_aux:
            with net.aux_scope():
                ops.conv1x1(x, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                            net.P[self.prefix + ".conv_shortcut.bias"], out,
                            a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", False, self.cout, self.cin, H * W, B))
        if ps:                                                # silu(gn(.)) written as the pre-split image its two consumers (convolution, weight gradient) read
            a1, m1, r1 = self.norm1.fwd_ps(x)
        else:
            a1 = torch.empty((B, self.cin, H, W), device=dev, dtype=torch.float32)
            m1, r1 = self.norm1.fwd(x, a1)
        h1 = torch.empty((B, self.cout, H, W), device=dev, dtype=torch.float32)
        self.conv1.fwd(a1, h1, rowadd=st.temb_all[:, self.temb_off:], rowadd_bstride=st.temb_all.stride(0))
        if ps:
            a2, m2, r2 = self.norm2.fwd_ps(h1)
        else:
            a2 = torch.empty_like(h1)
            m2, r2 = self.norm2.fwd(h1, a2)
        if self.has_sc:
            if sc_aux:
                net.aux_join()
            else:
                ops.conv1x1(x, net.P[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin),
                            net.P[self.prefix + ".conv_shortcut.bias"], out,
                            a_packed=_bx3_packed_1x1(net, self.prefix + ".conv_shortcut", False, self.cout, self.cin, H * W, B))
            self.conv2.fwd(a2, out, residual=out)
        else:
            self.conv2.fwd(a2, out, residual=x)
        if save:
            return (x, a1, m1, r1, h1, a2, m2, r2)
        return None

    def bwd(self, saved, dout, dx, st, dout_rs=None, extra2=None, dx_rs=None, dout_ps=None, dx_ps=None):
        """dout_rs: [B, cout] per-image channel sums of dout when its producer already made them (else a rowsum launch); extra2: a skip
        connection's gradient to add into dx; dx_rs: [B, cin] view to receive the sums of dx (both ride in the last GroupNorm backward).
        dout_ps: dout as a pre-split image when its producer wrote one too; dx_ps: receives dx as a pre-split image as well (the consumer of dx
        is another pre-split block's conv2)."""
        net = self.net
        x, a1, m1, r1, h1, a2, m2, r2 = saved
        B, _, H, W = x.shape
        dev = x.device
        if isinstance(a2, ops.PreSplit):
            return self._bwd_ps(saved, dout, dx, st, dout_rs, extra2, dx_rs, dout_ps, dx_ps)
        if a2 is None:                                        # folded-GroupNorm forward: the weight gradients' operands are made on their side stream
            a2 = self.norm2.fwd_later(h1)
            a1 = self.norm1.fwd_later(x)
        # conv2 (+ shortcut bias: both biases receive rowsum(dout))
        if dout_rs is None:
            bias_ws = net.scratch_bc(B, self.cout).view(B, self.cout)
            net.rowsum(dout, bias_ws)
        else:
            bias_ws = dout_rs
        dsc = None
        if self.has_sc and net.aux_fork():                    # shortcut input gradient (1x1, HBM-bound; needs dout only) beside the 3x3 chain
            with net.aux_scope():
                dsc = self._sc_dgrad(dout, B, H, W, dev)
        da2 = torch.empty((B, self.cout, H, W), device=dev, dtype=torch.float32)
        self.conv2.bwd(dout, a2, da2, bias_ws=bias_ws)
        dh1 = torch.empty_like(da2)
        # temb projection gradient rows + conv1 bias share rowsum(dh1)
        dt = st.d_temb_all[:, self.temb_off:self.temb_off + self.cout]
        if getattr(net, "fuse_gn_bwd", False):
            self.norm2.bwd(da2, h1, m2, r2, dh1, rowsum=dt)
        else:
            self.norm2.bwd(da2, h1, m2, r2, dh1)
            net.rowsum(dh1, dt, ws_ld=st.d_temb_all.stride(0))
        da1 = da2 if self.cin == self.cout else torch.empty((B, self.cin, H, W), device=dev, dtype=torch.float32)
        self.conv1.bwd(dh1, a1, da1, bias_ws=dt)
        if self.has_sc:
            net.wgrad(dout, x, net.G[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin), B_PLAIN,
                      math_mode=_wgrad1x1_math(net, dout, x))
            net.colsum_later(bias_ws, net.G[self.prefix + ".conv_shortcut.bias"], B, self.cout, ld=bias_ws.stride(0))
            if dsc is None:
                dsc = self._sc_dgrad(dout, B, H, W, dev)
            else:
                net.aux_join()                            # the shortcut's input gradient ran on the auxiliary stream beside the chain above
            self.norm1.bwd(da1, x, m1, r1, dx, extra=dsc, extra2=extra2, rowsum=dx_rs, dx_ps=dx_ps)
        else:
            self.norm1.bwd(da1, x, m1, r1, dx, extra=dout, extra2=extra2, rowsum=dx_rs, dx_ps=dx_ps)
        return dx

    def _bwd_ps(self, saved, dout, dx, st, dout_rs, extra2, dx_rs, dout_ps, dx_ps):
        """The same backward with pre-split operands (ps_plan): a1 / a2 are pre-split images from the forward pass, norm2's backward writes dh1 ONLY as a
        pre-split image (its two consumers are conv1's input and weight gradients), dout arrives pre-split from its producer or is packed on the
        weight-gradient stream, and norm1's backward can hand dx on pre-split as well."""
        net = self.net
        x, a1, m1, r1, h1, a2, m2, r2 = saved
        B, _, H, W = x.shape
        dev = x.device
        if dout_rs is None:
            bias_ws = net.scratch_bc(B, self.cout).view(B, self.cout)
            net.rowsum(dout, bias_ws)
        else:
            bias_ws = dout_rs
        dsc = None
        if self.has_sc and net.aux_fork():
            with net.aux_scope():
                dsc = self._sc_dgrad(dout, B, H, W, dev)
        da2 = torch.empty((B, self.cout, H, W), device=dev, dtype=torch.float32)
        self.conv2.bwd(dout, a2, da2, bias_ws=bias_ws, dout_ps=dout_ps)
        dh1 = ops.presplit_empty((B, self.cout, H, W), dev)
        dt = st.d_temb_all[:, self.temb_off:self.temb_off + self.cout]      # temb projection gradient rows + conv1 bias share rowsum(dh1)
        self.norm2.bwd(da2, h1, m2, r2, None, rowsum=dt, dx_ps=dh1)
        da1 = da2 if self.cin == self.cout else torch.empty((B, self.cin, H, W), device=dev, dtype=torch.float32)
        self.conv1.bwd(None, a1, da1, bias_ws=dt, dout_ps=dh1)
        if self.has_sc:
            net.wgrad(dout, x, net.G[self.prefix + ".conv_shortcut.weight"].view(self.cout, self.cin), B_PLAIN,
                      math_mode=_wgrad1x1_math(net, dout, x))
            net.colsum_later(bias_ws, net.G[self.prefix + ".conv_shortcut.bias"], B, self.cout, ld=bias_ws.stride(0))
            if dsc is None:
                dsc = self._sc_dgrad(dout, B, H, W, dev)
            else:
                net.aux_join()                            # the shortcut's input gradient ran on the auxiliary stream beside the chain above
            self.norm1.bwd(da1, x, m1, r1, dx, extra=dsc, extra2=extra2, rowsum=dx_rs, dx_ps=dx_ps)
        else:
            self.norm1.bwd(da1, x, m1, r1, dx, extra=dout, extra2=extra2, rowsum=dx_rs, dx_ps=dx_ps)
        return dx


class _Attn:
    """Spatial self-attention in NCHW (no transposes): see csrc/vd_attn.hip.  heads = ch // attention_head_dim (1 for
    the DDPM checkpoints); a head is a channel slice of q/k/v, so multi-head is the same GEMMs on offset views."""

    def __init__(self, net, prefix, ch, head_dim=None, groups=None):
        self.net, self.prefix, self.ch = net, prefix, ch
        self.heads = 1 if head_dim is None else ch // head_dim
        if ch % self.heads:
            raise ValueError(f"{prefix}: {ch} channels do not split into heads of {head_dim}")
        self.norm = _Norm(net, prefix + ".group_norm", ch, False, groups)
        self.qkv_w, self.qkv_b = net._decl_qkv(prefix, ch)
        net._decl(prefix + ".to_out.0.weight", (ch, ch), fan_in=ch)
        net._decl(prefix + ".to_out.0.bias", (ch,), fan_in=ch, is_bias=True)
        self.scale = 1.0 / math.sqrt(ch // self.heads)

    def fwd(self, x, out, st, save):
        net, Cc = self.net, self.ch
        B, _, H, W = x.shape
        N = H * W
        dev = x.device
        g = torch.empty((B, Cc, H, W), device=dev, dtype=torch.float32)
        mean, rstd = self.norm.fwd(x, g)
        qkv = torch.empty((B, 3 * Cc, H, W), device=dev, dtype=torch.float32)
        ops.conv1x1(g, net.Pq[self.qkv_w], net.Pq[self.qkv_b], qkv, a_packed=_bx3_packed_1x1(net, self.prefix + "::qkv", False, 3 * Cc, Cc, N, B))
        o = torch.empty((B, Cc, H, W), device=dev, dtype=torch.float32)
        nh, dh = self.heads, Cc // self.heads
        if _split(net) and getattr(net, "fused_attention", False) and ops.attn_flash_eligible(nh, dh, N):
            # more than 256 tokens per head (config #5's 32x32 level): online softmax over key blocks, only lse [B, heads, N] is saved
            lse = torch.empty((B, nh, N), device=dev, dtype=torch.float32) if save else None
            ops.attn_flash_fwd(qkv, o, lse, nh, dh, N, self.scale)
            ops.conv1x1(o, net.P[self.prefix + ".to_out.0.weight"], net.P[self.prefix + ".to_out.0.bias"], out, residual=x,
                        a_packed=_bx3_packed_1x1(net, self.prefix + ".to_out.0", False, Cc, Cc, N, B))
            return (x, mean, rstd, g, qkv, lse, o) if save else None
        if _split(net) and getattr(net, "fused_attention", False) and ops.attn_core_eligible(nh, dh, N):
            # one launch: scores and probabilities stay in registers; P reaches HBM only when a backward pass will read it
            P = torch.empty((B, nh, N, N), device=dev, dtype=torch.float32) if save else None
            ops.attn_core_fwd(qkv, o, P, nh, dh, N, self.scale)
            ops.conv1x1(o, net.P[self.prefix + ".to_out.0.weight"], net.P[self.prefix + ".to_out.0.bias"], out, residual=x,
                        a_packed=_bx3_packed_1x1(net, self.prefix + ".to_out.0", False, Cc, Cc, N, B))
            return (x, mean, rstd, g, qkv, P, o) if save else None
        P = torch.empty((B, nh, N, N), device=dev, dtype=torch.float32)
        if nh > 1:
            if N % 64:
                raise NotImplementedError(f"multi-head attention needs a multiple of 64 tokens per image (got {H}x{W})")
            # all heads in one launch: batch item (b, h) -> b * 3C*N + h * dh*N inside qkv (two-level batch of vd_gemm)
            bs, hs = 3 * Cc * N, dh * N
            q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
            two = dict(nb2=nh, NP=N)
            ops.gemm(k, q, P, M=N, N=B * nh * N, K=dh, a_mode=A_COL, b_mode=B_PLAIN, lda=N, a_bstride=bs, a_b2stride=hs, ldb=N,
                     b_bstride=bs, b_b2stride=hs, ldd=N, d_bstride=nh * N * N, d_b2stride=N * N, alpha=self.scale, **two, math_mode=_amath(net, N, dh, N))
            ops.softmax_col_fwd(P, B * nh, N)
            ops.gemm(v, P, o, M=dh, N=B * nh * N, K=N, a_mode=A_ROW, b_mode=B_PLAIN, lda=N, a_bstride=bs, a_b2stride=hs, ldb=N,
                     b_bstride=nh * N * N, b_b2stride=N * N, ldd=N, d_bstride=Cc * N, d_b2stride=hs, **two, math_mode=_amath(net, dh, N, N))
        elif N < 64 or (N == 64 and B >= 64):     # one workgroup per image: needs a batch that fills the chip
            ops.attn_small_fwd(qkv, o, P, Cc, N, self.scale)
        else:
            q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
            bs = 3 * Cc * N
            # St[j][i] = scale * sum_c k[c][j] q[c][i]
            ops.gemm(k, q, P, M=N, N=B * N, K=Cc, a_mode=A_COL, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N,
                     b_bstride=bs, ldd=N, d_bstride=N * N, alpha=self.scale, math_mode=_amath(net, N, Cc, N))
            ops.softmax_col_fwd(P, B, N)
            # o[c][i] = sum_j v[c][j] P[j][i]
            ops.gemm(v, P, o, M=Cc, N=B * N, K=N, a_mode=A_ROW, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N,
                     b_bstride=N * N, ldd=N, d_bstride=Cc * N, math_mode=_amath(net, Cc, N, N))
        ops.conv1x1(o, net.P[self.prefix + ".to_out.0.weight"], net.P[self.prefix + ".to_out.0.bias"], out, residual=x,
                    a_packed=_bx3_packed_1x1(net, self.prefix + ".to_out.0", False, Cc, Cc, N, B))
        if save:
            return (x, mean, rstd, g, qkv, P, o)
        return None

    def bwd(self, saved, dout, dx, st, dout_rs=None, extra2=None, dx_rs=None, dout_ps=None, dx_ps=None):
        """(dout_ps is not used: the 1x1 products read f32; dx_ps: dx also as a pre-split image for a pre-split ResnetBlock that consumes it.)"""
        net, Cc = self.net, self.ch
        x, mean, rstd, g, qkv, P, o = saved
        B, _, H, W = x.shape
        N = H * W
        dev = x.device
        wo = net.P[self.prefix + ".to_out.0.weight"]
        net.wgrad(dout, o, net.G[self.prefix + ".to_out.0.weight"], B_PLAIN, math_mode=_wgrad1x1_math(net, dout, o))
        if dout_rs is None:
            bias_ws = net.scratch_bc(B, Cc).view(B, Cc)
            net.rowsum(dout, bias_ws)
        else:
            bias_ws = dout_rs
        net.colsum_later(bias_ws, net.G[self.prefix + ".to_out.0.bias"], B, Cc, ld=bias_ws.stride(0))
        do = torch.empty((B, Cc, H, W), device=dev, dtype=torch.float32)
        ops.gemm(wo, dout, do, M=Cc, N=B * N, K=Cc, a_mode=A_COL, b_mode=B_PLAIN, NP=N, lda=Cc, ldb=N,
                 b_bstride=ops._img(dout)[4], ldd=N, d_bstride=Cc * N, a_packed=_bx3_packed_1x1(net, self.prefix + ".to_out.0", True, Cc, Cc, N, B))
        dqkv = torch.empty((B, 3 * Cc, H, W), device=dev, dtype=torch.float32)
        nh, dh = self.heads, Cc // self.heads
        if P.dim() == 3:                                   # saved by the flash forward: P is lse [B, heads, N]
            ops.attn_flash_bwd(qkv, o, do, P, dqkv, nh, dh, N, self.scale)
        elif _split(net) and getattr(net, "fused_attention", False) and ops.attn_core_eligible(nh, dh, N):
            # dP, the softmax gradient and dq in one launch (dP never reaches HBM); dv and dk are products of the saved P / of dS
            dS = torch.empty((B, nh, N, N), device=dev, dtype=torch.float32)
            ops.attn_core_bwd(qkv, P, o, do, dS, dqkv, nh, dh, N, self.scale)
            bs, hs, pbs, NN = 3 * Cc * N, dh * N, nh * N * N, N * N
            q = qkv[:, :Cc]
            dk, dv = dqkv[:, Cc:2 * Cc], dqkv[:, 2 * Cc:]
            two = dict(nb2=nh, NP=N, N=B * nh * N) if nh > 1 else dict(NP=N, N=B * N)
            st2 = (lambda a, b: dict(a_b2stride=a, b_b2stride=b, d_b2stride=hs)) if nh > 1 else (lambda a, b: {})
            ops.gemm(do, P, dv, M=dh, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, lda=N, a_bstride=Cc * N, ldb=N, b_bstride=pbs, ldd=N,
                     d_bstride=bs, **two, **st2(hs, NN), math_mode=_amath(net, dh, N, N))
            ops.gemm(q, dS, dk, M=dh, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, lda=N, a_bstride=bs, ldb=N, b_bstride=pbs, ldd=N,
                     d_bstride=bs, **two, **st2(hs, NN), math_mode=_amath(net, dh, N, N))
        elif nh > 1:
            bs, hs, pbs, NN = 3 * Cc * N, dh * N, nh * N * N, N * N
            dP = torch.empty((B, nh, N, N), device=dev, dtype=torch.float32)
            q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
            dq, dk, dv = dqkv[:, :Cc], dqkv[:, Cc:2 * Cc], dqkv[:, 2 * Cc:]
            two = dict(nb2=nh, NP=N, N=B * nh * N)
            # dv[c][j] = sum_i do[c][i] P[j][i]
            ops.gemm(do, P, dv, M=dh, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, lda=N, a_bstride=Cc * N, a_b2stride=hs, ldb=N,
                     b_bstride=pbs, b_b2stride=NN, ldd=N, d_bstride=bs, d_b2stride=hs, **two, math_mode=_amath(net, dh, N, N))
            # dP[j][i] = sum_c v[c][j] do[c][i]
            ops.gemm(v, do, dP, M=N, K=dh, a_mode=A_COL, b_mode=B_PLAIN, lda=N, a_bstride=bs, a_b2stride=hs, ldb=N,
                     b_bstride=Cc * N, b_b2stride=hs, ldd=N, d_bstride=pbs, d_b2stride=NN, **two, math_mode=_amath(net, N, dh, N))
            ops.softmax_col_bwd(P, dP, B * nh, N, self.scale)
            # dq[c][i] = sum_j k[c][j] dS[j][i] ;  dk[c][j] = sum_i q[c][i] dS[j][i]
            ops.gemm(k, dP, dq, M=dh, K=N, a_mode=A_ROW, b_mode=B_PLAIN, lda=N, a_bstride=bs, a_b2stride=hs, ldb=N,
                     b_bstride=pbs, b_b2stride=NN, ldd=N, d_bstride=bs, d_b2stride=hs, **two, math_mode=_amath(net, dh, N, N))
            ops.gemm(q, dP, dk, M=dh, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, lda=N, a_bstride=bs, a_b2stride=hs, ldb=N,
                     b_bstride=pbs, b_b2stride=NN, ldd=N, d_bstride=bs, d_b2stride=hs, **two, math_mode=_amath(net, dh, N, N))
        elif N < 64 or (N == 64 and B >= 64):
            ops.attn_small_bwd(qkv, P, do, dqkv, Cc, N, self.scale)
        else:
            q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
            dq, dk, dv = dqkv[:, :Cc], dqkv[:, Cc:2 * Cc], dqkv[:, 2 * Cc:]
            bs = 3 * Cc * N
            # dv[c][j] = sum_i do[c][i] P[j][i]
            ops.gemm(do, P, dv, M=Cc, N=B * N, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, NP=N, lda=N, a_bstride=Cc * N, ldb=N,
                     b_bstride=N * N, ldd=N, d_bstride=bs, math_mode=_amath(net, Cc, N, N))
            # dP[j][i] = sum_c v[c][j] do[c][i]
            dP = torch.empty((B, N, N), device=dev, dtype=torch.float32)
            ops.gemm(v, do, dP, M=N, N=B * N, K=Cc, a_mode=A_COL, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N,
                     b_bstride=Cc * N, ldd=N, d_bstride=N * N, math_mode=_amath(net, N, Cc, N))
            ops.softmax_col_bwd(P, dP, B, N, self.scale)           # dP -> dS (scaled)
            # dq[c][i] = sum_j k[c][j] dS[j][i]
            ops.gemm(k, dP, dq, M=Cc, N=B * N, K=N, a_mode=A_ROW, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N,
                     b_bstride=N * N, ldd=N, d_bstride=bs, math_mode=_amath(net, Cc, N, N))
            # dk[c][j] = sum_i q[c][i] dS[j][i]
            ops.gemm(q, dP, dk, M=Cc, N=B * N, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, NP=N, lda=N, a_bstride=bs, ldb=N,
                     b_bstride=N * N, ldd=N, d_bstride=bs, math_mode=_amath(net, Cc, N, N))
        net.wgrad(dqkv, g, net.Gq[self.qkv_w], B_PLAIN, math_mode=_wgrad1x1_math(net, dqkv, g))
        ws3 = net.scratch_bc(B, 3 * Cc)
        net.rowsum(dqkv, ws3)
        net.colsum_later(ws3, net.Gq[self.qkv_b], B, 3 * Cc)
        dg = torch.empty((B, Cc, H, W), device=dev, dtype=torch.float32)
        ops.gemm(net.Pq[self.qkv_w], dqkv, dg, M=Cc, N=B * N, K=3 * Cc, a_mode=A_COL, b_mode=B_PLAIN, NP=N, lda=Cc, ldb=N,
                 b_bstride=3 * Cc * N, ldd=N, d_bstride=Cc * N, a_packed=_bx3_packed_1x1(net, self.prefix + "::qkv", True, Cc, 3 * Cc, N, B))
        self.norm.bwd(dg, x, mean, rstd, dx, extra=dout, extra2=extra2, rowsum=dx_rs, dx_ps=dx_ps)
        return dx


# ----------------------------------------------------------------------------------------------------------- network
class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, t, anchor):
        out, st = net._run_forward(x, t, save=True)
        ctx.net, ctx.st = net, st
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.net._run_backward(ctx.st, dout.contiguous())
        ctx.st = None
        return None, None, None, None


class UNet2DModel(nn.Module):
    """Drop-in for ``diffusers.UNet2DModel`` (positional time embedding; DownBlock2D/AttnDownBlock2D/UpBlock2D/
    AttnUpBlock2D) on MI355X."""

    def __init__(self, in_channels=3, out_channels=3, sample_size=32, block_out_channels=(128, 256, 256, 256),
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 layers_per_block=2, norm_num_groups=32, norm_eps=1e-6, downsample_padding=0, flip_sin_to_cos=False,
                 freq_shift=1, attention_head_dim=None, act_fn="silu", time_embedding_type="positional",
                 center_input_sample=False, mid_block_scale_factor=1, device=None, **unused):
        super().__init__()
        if time_embedding_type != "positional" or act_fn != "silu" or downsample_padding not in (0, 1) or center_input_sample:
            raise NotImplementedError("only the DDPM-style UNet2DModel configuration is implemented natively")
        for t in tuple(down_block_types) + tuple(up_block_types):
            if t not in ("DownBlock2D", "AttnDownBlock2D", "UpBlock2D", "AttnUpBlock2D"):
                raise NotImplementedError(f"block type {t}")
        boc = list(block_out_channels)
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, sample_size=sample_size, block_out_channels=tuple(boc),
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            layers_per_block=layers_per_block, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
            downsample_padding=downsample_padding, flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift,
            attention_head_dim=attention_head_dim, act_fn=act_fn, time_embedding_type=time_embedding_type,
            center_input_sample=center_input_sample, mid_block_scale_factor=mid_block_scale_factor)
        self.in_channels, self.out_channels, self.sample_size = in_channels, out_channels, sample_size
        self.groups, self.eps = norm_num_groups, norm_eps
        if device is not None:
            self._dev = torch.device(device)
        elif torch.cuda.is_available():
            self._dev = torch.device("cuda", torch.cuda.current_device())
        else:       # structure-only use (state-dict surgery, tests): any compute call still fails loudly in lib.require_device()
            self._dev = torch.device("cpu")

        # ---- declare parameters (order = layout in the flat buffer) ----
        self._decls: List[Tuple[str, Tuple[int, ...], dict]] = []
        self._temb: List[Tuple[str, int]] = []          # (prefix, cout) in execution order
        self._qkv: List[Tuple[str, int]] = []
        temb_dim = boc[0] * 4
        self.time_dim0, self.temb_dim = boc[0], temb_dim
        self._decl("time_embedding.linear_1.weight", (temb_dim, boc[0]), fan_in=boc[0])
        self._decl("time_embedding.linear_1.bias", (temb_dim,), fan_in=boc[0], is_bias=True)
        self._decl("time_embedding.linear_2.weight", (temb_dim, temb_dim), fan_in=temb_dim)
        self._decl("time_embedding.linear_2.bias", (temb_dim,), fan_in=temb_dim, is_bias=True)
        self._conv_in = _Conv(self, "conv_in", in_channels, boc[0])

        self.down: List[dict] = []
        ch = boc[0]
        skip_ch = [ch]
        for i, typ in enumerate(down_block_types):
            cin, ch = ch, boc[i]
            blk = {"res": [], "attn": [], "ds": None}
            for j in range(layers_per_block):
                blk["res"].append(_Resnet(self, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else ch, ch))
                if typ == "AttnDownBlock2D":
                    blk["attn"].append(_Attn(self, f"down_blocks.{i}.attentions.{j}", ch, attention_head_dim))
                skip_ch.append(ch)
            if i != len(boc) - 1:
                blk["ds"] = _Conv(self, f"down_blocks.{i}.downsamplers.0.conv", ch, ch, mode=B_CONV3_S2, pad=downsample_padding)
                skip_ch.append(ch)
            self.down.append(blk)
        self.mid_res = [_Resnet(self, "mid_block.resnets.0", ch, ch), _Resnet(self, "mid_block.resnets.1", ch, ch)]
        self.mid_attn = _Attn(self, "mid_block.attentions.0", ch, attention_head_dim)

        rev = boc[::-1]
        self.up: List[dict] = []
        out_ch = rev[0]
        sk = list(skip_ch)
        for i, typ in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            blk = {"res": [], "attn": [], "us": None, "h_ch": [], "skip_ch": []}
            for j in range(layers_per_block + 1):
                s_ch = sk.pop()
                assert s_ch == (in_ch if j == layers_per_block else out_ch)
                h_ch = prev if j == 0 else out_ch
                blk["res"].append(_Resnet(self, f"up_blocks.{i}.resnets.{j}", h_ch + s_ch, out_ch))
                blk["h_ch"].append(h_ch)
                blk["skip_ch"].append(s_ch)
                if typ == "AttnUpBlock2D":
                    blk["attn"].append(_Attn(self, f"up_blocks.{i}.attentions.{j}", out_ch, attention_head_dim))
            if i != len(boc) - 1:
                blk["us"] = _Conv(self, f"up_blocks.{i}.upsamplers.0.conv", out_ch, out_ch, mode=B_CONV3_UP)
            self.up.append(blk)
        assert not sk
        self.norm_out = _Norm(self, "conv_norm_out", boc[0], True)
        self._conv_out = _Conv(self, "conv_out", boc[0], out_channels)
        self._materialise()

    # ------------------------------------------------------------------------------------------ parameter plumbing
    def _decl(self, name, shape, fan_in=None, is_bias=False, ones=False, zeros=False):
        self._decls.append((name, tuple(shape), dict(fan_in=fan_in, is_bias=is_bias, ones=ones, zeros=zeros)))

    def _decl_temb(self, prefix, cout) -> int:
        off = sum(c for _, c in self._temb)
        self._temb.append((prefix, cout))
        return off

    def _decl_qkv(self, prefix, ch):
        self._qkv.append((prefix, ch))
        return prefix + "::qkv_w", prefix + "::qkv_b"

    def _materialise(self):
        # layout: [temb weights][temb biases][per attention: q,k,v weights][q,k,v biases][everything else]
        layout: List[Tuple[str, Tuple[int, ...], dict]] = []
        for prefix, cout in self._temb:
            layout.append((prefix + ".weight", (cout, self.temb_dim), dict(fan_in=self.temb_dim)))
        for prefix, cout in self._temb:
            layout.append((prefix + ".bias", (cout,), dict(fan_in=self.temb_dim, is_bias=True)))
        for prefix, ch in self._qkv:
            for n in ("to_q", "to_k", "to_v"):
                layout.append((f"{prefix}.{n}.weight", (ch, ch), dict(fan_in=ch)))
            for n in ("to_q", "to_k", "to_v"):
                layout.append((f"{prefix}.{n}.bias", (ch,), dict(fan_in=ch, is_bias=True)))
        layout.extend(self._decls)
        offs, total = {}, 0
        for name, shape, _ in layout:
            n = int(math.prod(shape))
            offs[name] = (total, n, shape)
            total += (n + 3) // 4 * 4           # keep every parameter 16-byte aligned
        self._layout, self._offs, self.flat_numel = layout, offs, total
        dev = self._dev
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        for name, shape, _ in layout:
            off, n, _ = offs[name]
            parts = name.split(".")
            holder = _ensure_path(self, parts[:-1])
            p = nn.Parameter(self.flat_param[off:off + n].view(shape), requires_grad=True)
            p.grad = self.flat_grad[off:off + n].view(shape)
            holder.register_parameter(parts[-1], p)
            self.P[name], self.G[name] = p.data, p.grad
        # fused views
        n_t = sum(c for _, c in self._temb)
        o0 = offs[self._temb[0][0] + ".weight"][0]
        ob = offs[self._temb[0][0] + ".bias"][0]
        self.temb_total = n_t
        self.Wt_all = self.flat_param[o0:o0 + n_t * self.temb_dim].view(n_t, self.temb_dim)
        self.bt_all = self.flat_param[ob:ob + n_t]
        self.gWt_all = self.flat_grad[o0:o0 + n_t * self.temb_dim].view(n_t, self.temb_dim)
        self.gbt_all = self.flat_grad[ob:ob + n_t]
        self.Pq, self.Gq = {}, {}
        for prefix, ch in self._qkv:
            ow, obq = offs[f"{prefix}.to_q.weight"][0], offs[f"{prefix}.to_q.bias"][0]
            for src, dst in ((self.flat_param, self.Pq), (self.flat_grad, self.Gq)):
                dst[prefix + "::qkv_w"] = src[ow:ow + 3 * ch * ch].view(3 * ch, ch)
                dst[prefix + "::qkv_b"] = src[obq:obq + 3 * ch]
        # transposed conv weights for dgrad, refreshed once per backward
        self._wt_offs, wt_total = {}, 0
        for name, shape, _ in layout:
            if name.endswith(".weight") and len(shape) == 4 and shape[2] == 3 and name != "conv_in.weight":
                self._wt_offs[name[:-7]] = wt_total
                wt_total += int(math.prod(shape))
        self._wt_total = wt_total
        self._wt_buf: Optional[torch.Tensor] = None
        self.wgrad_ws: Optional[torch.Tensor] = None
        self._cs_pool: Optional[torch.Tensor] = None
        self._cs_tables: Dict[tuple, torch.Tensor] = {}
        self._cs_cols_hint = 4 * sum(int(math.prod(sh)) for _, sh, _ in layout if len(sh) == 1) + 4096
        self._cs_off, self._cs_jobs, self._cs_B, self._cs_retired = 0, [], 0, []
        self._d_temb: Dict[int, torch.Tensor] = {}
        half = self.time_dim0 // 2
        # [UPSTREAM] get_timestep_embedding: exponent = -ln(1e4) * arange(half) / (half - freq_shift), fp32 torch ops
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
        exponent = exponent / (half - self.config.freq_shift)
        self.freqs = torch.exp(exponent).to(dev)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        # Gradient buckets in the order backward completes them (flat-buffer ranges): [up blocks | out], [mid], [conv_in |
        # down blocks], [time-embedding MLP, all time_emb_proj, all q/k/v].  bucket_ready_hook(i) is called from the
        # explicit backward as soon as bucket i is final, so a trainer can overlap its all-reduce with the rest.
        def _late(k):     # tensors laid out in the fused head of the buffer (time_emb_proj / q,k,v): bucket 3
            return ".time_emb_proj." in k or ".to_q." in k or ".to_k." in k or ".to_v." in k
        o_in = offs["conv_in.weight"][0]
        o_mid = min(v[0] for k, v in offs.items() if k.startswith("mid_block.") and not _late(k))
        o_up = min(v[0] for k, v in offs.items() if k.startswith("up_blocks.") and not _late(k))
        self.grad_buckets = [(o_up, total), (o_mid, o_up), (o_in, o_mid), (0, o_in)]
        self.bucket_ready_hook = None
        # no-grad forward: fold GroupNorm + SiLU into the 3x3 convolutions' loaders (vd_gemm gn_ss).  Measured on MI355X it does NOT
        # pay on the exact-f32 kernels: the transform sits in the store phase of the K-step (conv 439 -> 455 us, 373 -> 393 us) and costs
        # more than the saved normalise pass (26 -> 14 us): 8.22 vs 8.30 img/s for DDPM-1000.  On the split-precision kernels the patch
        # is stored once per 16 channels x 9 taps and the fold pays (7.39 -> 7.22 ms per sampler step).  None = on for "bf16x3" only.
        self.fuse_gn_inference = None
        # attention blocks of 256 tokens run as ONE fused launch per direction in the split-precision arithmetic (vd_attn_core_*);
        # False keeps the three-launch GEMM / column-softmax / GEMM sequence (the only path of the exact-f32 arithmetic)
        self.fused_attention = os.environ.get("VILLAN_FUSED_ATTENTION", "1") != "0"
        # sampler loops replay the no-grad forward from a HIP graph captured once per batch shape (pipelines.GraphedForward)
        self.sampler_graph = os.environ.get("VILLAN_SAMPLER_GRAPH", "1") != "0"
        # weight gradients of a gradient bucket run as grouped launches (see wgrad()); False: one launch pair per convolution
        self.group_wgrad = os.environ.get("VILLAN_GROUP_WGRAD", "1") != "0"
        self._wg_jobs: Dict[int, list] = {}
        # ... on a side stream, flushed every `wgrad_flush_jobs` queued convolutions (0: only when a gradient bucket completes)
        self.wgrad_stream = os.environ.get("VILLAN_WGRAD_STREAM", "1") != "0"
        self.wgrad_flush_jobs = int(os.environ.get("VILLAN_WGRAD_FLUSH_JOBS", "24"))
        self._wg_side, self._wg_keep, self._rs_jobs = None, [], []
        # bias-gradient row sums and skip-connection gradient adds ride in the GroupNorm backward that writes the tensor they read
        # (vd_groupnorm_bwd_fused) instead of ~50 rowsum + 12 add_strided launches per step; False: the separate launches
        self.fuse_gn_bwd = os.environ.get("VILLAN_FUSE_GN_BWD", "1") != "0"
        # round 5: GroupNorm forward / backward write PRE-SPLIT bf16 (hi, lo) images for the 3x3 convolutions and their weight gradients
        # (csrc/vd_presplit.hip; _Resnet.ps_plan decides per block); VILLAN_PRESPLIT=0: round 4's converting kernels
        self.presplit = os.environ.get("VILLAN_PRESPLIT", "1") != "0"
        self.nograd_presplit = os.environ.get("VILLAN_NOGRAD_PRESPLIT", "1") != "0"
        self.us_dgrad_presplit = os.environ.get("VILLAN_US_DGRAD_PRESPLIT", "1") != "0"
        self.us_fwd_presplit = os.environ.get("VILLAN_US_FWD_PRESPLIT", "1") != "0"
        # round 5: the 1x1 shortcut of a ResnetBlock (forward) and its input gradient (backward) on an auxiliary stream beside the block's 3x3
        # chain (VILLAN_SC_STREAM=0: in line)
        self.sc_stream = os.environ.get("VILLAN_SC_STREAM", "1") != "0"
        self._aux_stream = None
        # no-grad forward: the statistics of a ResnetBlock2D's second GroupNorm are summed in the first convolution's epilogue
        # (vd_gemm_desc.gn_part) instead of a read of its output; False: the statistics pass
        self.gn_stats_in_epilogue = os.environ.get("VILLAN_GN_STATS_IN_EPILOGUE", "1") != "0"
        # opt-in (round 4, measured and NOT the default): training forward with GroupNorm + SiLU folded into the 16x16 / 32x32 convolutions'
        # loaders; silu(gn(.)) for the weight gradients is recomputed on the side stream in the backward pass.  Bit-identical results, 1.7 GB
        # less saved activations at B = 128, but 18.07 -> 18.44 ms per step (profiles/r04_gn_defer_ab.txt): the statistics pass plus the
        # recomputation move MORE bytes than the normalise pass they replace, and on a power-bound chip an HBM-bound pass running beside the
        # MFMA-bound kernels is not free (the clock drops for both)
        self.defer_gn_fwd = os.environ.get("VILLAN_DEFER_GN_FWD", "0") != "0"
        # round 4: training forward of the 16x16 / 32x32 resnets without normalise passes -- the convolution's GroupNorm-folding loader writes
        # silu(gn(x)) as a side output, norm2's statistics come from conv1's epilogue.  Opt-in: measured +0.2 ms / step (profiles/r04_gn_actout_ab.txt)
        self.fold_gn_train = os.environ.get("VILLAN_FOLD_GN_TRAIN", "0") != "0"
        self._gn_jobs = []
        self._pk_jobs = []
        # "bf16x3": eligible 3x3 convolutions (forward and stride-1 input gradient at 8x8 / 16x16 / 32x32) run on the bf16 matrix
        # cores as hi*hi + hi*lo + lo*hi with f32 accumulation (~1e-5 of the exact result); "f32": everything on the exact f32 MFMA.
        self.conv_math = CONV_MATH_DEFAULT
        self._packed: Optional[_PackedConvWeights] = None
        self._packed16: Optional[_PackedConvWeights] = None     # f16 operands of the opt-in mixed-precision mode (conv_math = "f16")
        self._wt_fresh = set()
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None):
        """torch default init of Conv2d / Linear (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in))) and GroupNorm."""
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        host = torch.zeros(self.flat_numel, dtype=torch.float32)
        for name, shape, meta in self._layout:
            off, n, _ = self._offs[name]
            if meta.get("ones"):
                host[off:off + n] = 1.0
            elif meta.get("zeros"):
                host[off:off + n] = 0.0
            else:
                bound = 1.0 / math.sqrt(meta["fan_in"])
                host[off:off + n] = (torch.rand(n, generator=gen) * 2 - 1) * bound
        self.flat_param.copy_(host)

    def zero_grad(self, set_to_none: bool = False):          # keeps .grad views into the flat buffer
        ops.scale_(self.flat_grad, 0.0)

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {}
        for k, v in state_dict.items():
            parts = k.split(".")
            if "attentions" in parts and len(parts) >= 2 and parts[-2] in LEGACY_ATTN:   # diffusers < 0.17 names
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
                    assert tuple(v.shape) == tuple(shape) or v.numel() == n, (k, v.shape, shape)
                    self.flat_param[off:off + n].copy_(v.reshape(-1).to(torch.float32))
        self.weights_changed()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def refresh_packed(self, bwd: bool):
        """Rebuild whatever packed operand sets exist (split-precision, f16) for the current weights -- what a captured graph's caller does before
        a replay (the graph reads the packed buffers, it does not rebuild them)."""
        for pk in (self._packed, self._packed16):
            if pk is not None:
                pk.refresh(bwd)

    def weights_changed(self):
        """Drop the packed split-precision operands.  They are rebuilt automatically after optimiser steps, `load_state_dict`,
        `reset_parameters` and any in-place torch op on a parameter (keyed on the flat buffer's version counter); call this after
        writing parameters behind autograd's back (`p.data.copy_(...)`, raw pointers), which no counter sees."""
        for pk in (self._packed, self._packed16):
            if pk is not None:
                pk.key = {False: None, True: None}

    def to(self, *args, **kwargs):          # parameters are views of one flat device buffer: never re-materialise
        return self

    def cuda(self, device=None):
        return self

    @property
    def device(self):
        return self._dev

    @property
    def dtype(self):
        return torch.float32

    # ------------------------------------------------------------------------------------------ scratch
    # ---- per-batch partial sums of the bias / GroupNorm-parameter gradients and their deferred reduction ----
    # Every [B, C] partial gets its own region of one pool so that the ~180 column sums of a backward pass can be reduced by
    # ONE vd_colsum_segmented launch per gradient bucket instead of a 5 us launch each.
    def _cs_begin(self, B):
        need = B * self._cs_cols_hint
        if self._cs_pool is None or self._cs_pool.numel() < need:
            self._cs_pool = torch.empty(need, device=self._dev, dtype=torch.float32)
            self._cs_tables = {}
        self._cs_off, self._cs_jobs, self._cs_B = 0, [], B
        self._cs_retired = []                                     # pools outgrown during this pass: row sums handed from layer to layer may still live there

    def scratch_bc(self, B, Cc, slot=0):
        n = (B * Cc + 3) // 4 * 4
        if self._cs_off + n > self._cs_pool.numel():             # first pass with an under-estimated hint: flush and grow
            self._cs_flush()
            self._cs_retired.append(self._cs_pool)
            self._cs_cols_hint *= 2
            self._cs_pool = torch.empty(B * self._cs_cols_hint, device=self._dev, dtype=torch.float32)
            self._cs_tables, self._cs_off = {}, 0
        t = self._cs_pool[self._cs_off:self._cs_off + B * Cc]
        self._cs_off += n
        return t

    def colsum_later(self, ws, out, B, Cc, ld=None):
        """out[c] += sum_b ws[b*ld + c], executed at the next _cs_flush() (ws must stay untouched until then)."""
        ld = Cc if ld is None else ld
        wp, op = ws.data_ptr(), out.data_ptr()
        for c0 in range(0, Cc, 64):
            self._cs_jobs.append((wp + 4 * c0, op + 4 * c0, min(64, Cc - c0), ld))

    # ---- weight gradients: deferred and launched GROUPED per kernel class (ops.conv_wgrad_group) ----
    # A weight gradient has a small output and a reduction over batch * pixels: launched alone it must split that reduction ~32 ways
    # to fill the chip and moves 32 partial copies of dW through memory.  The gradients of one bucket are independent of each other
    # and off the critical path of the backward pass, so they are queued (operands kept alive) and run as a few grouped launches when
    # the bucket is final.
    def wgrad(self, dy, x, dw2d, mode, pad=0, math_mode=0):
        if math_mode == 1 and self.group_wgrad:
            d = ops.wgrad_desc(dy, x, dw2d, mode, None, accumulate=True, pad=pad, math_mode=1)
            cls = ops.wgrad_group_class(d)
            if cls:
                self._wg_jobs.setdefault(cls, []).append((d, dy, x))
                if self.wgrad_stream and self.wgrad_flush_jobs and sum(len(v) for v in self._wg_jobs.values()) >= self.wgrad_flush_jobs:
                    self._wg_flush()
                return
        if self._gn_jobs:                                     # an ungrouped weight gradient runs NOW: its operand may be a deferred silu(gn(.))
            for norm, xx, yy in self._gn_jobs:
                norm._fwd_now(xx, yy)
            self._gn_jobs = []
        ops.conv_wgrad(dy, x, dw2d, mode, self.wgrad_ws, accumulate=True, pad=pad, math_mode=math_mode)

    # ---- auxiliary stream: independent HBM-bound launches (the 1x1 shortcut and its input gradient) beside the 3x3 chain of a ResnetBlock ----
    def aux_fork(self) -> bool:
        """Make the auxiliary stream wait for everything issued on the current stream so far; False when the overlap is switched off."""
        if not self.sc_stream or self._dev.type != "cuda":
            return False
        if self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream(device=self._dev)
        # the packed weight operands are rebuilt lazily by their first user ON THAT USER'S STREAM: make sure that is this one, before the fork
        # (a rebuild issued from the auxiliary stream would race with the main stream's convolutions reading the same buffer)
        if _split(self) and self._packed is None:
            self._packed = _PackedConvWeights(self)
        if self.conv_math == "f16" and self._packed16 is None:
            self._packed16 = _PackedConvWeights(self, f16=True)
        for pk in (self._packed, self._packed16):
            if pk is not None:
                pk.refresh(False)
                pk.refresh(True)
        self._aux_stream.wait_stream(torch.cuda.current_stream(self._dev))
        return True

    def aux_scope(self):
        import contextlib
        stack = contextlib.ExitStack()
        stack.enter_context(torch.cuda.stream(self._aux_stream))
        stack.enter_context(ops.ws_slot(ops.AUX_WS_SLOT))       # its own split-K workspace slot (never one of the sampler streams')
        return stack

    def aux_join(self):
        torch.cuda.current_stream(self._dev).wait_stream(self._aux_stream)

    def pack_later(self, t):
        """The pre-split image of `t` (a dY whose producer wrote f32 only) for a queued pre-split weight gradient: packed on the weight-gradient
        stream right before the grouped launch that reads it (off the critical path), or now when there is no such stream."""
        out = ops.presplit_empty(t.shape, t.device)
        if self.wgrad_stream and self.group_wgrad:
            self._pk_jobs.append((t, out))
        else:
            ops.presplit_pack(t, out=out)
        return out

    def gn_later(self, norm, x, y):
        """y = norm(x) (+SiLU) before the next grouped weight-gradient launch, on its stream (see _Norm.fwd_later)."""
        if self.wgrad_stream and self.group_wgrad:
            self._gn_jobs.append((norm, x, y))
        else:
            norm._fwd_now(x, y)

    def rowsum(self, x, ws, ws_ld=None):
        """Bias-gradient partials ws[b][m] = sum_p x[b][m][p]: consumed only when the bucket is flushed, so with the side stream they
        ride there too (x is a dY of a queued weight gradient or is kept referenced like one)."""
        if self.wgrad_stream and self.group_wgrad:
            self._rs_jobs.append((x, ws, ws_ld))
        else:
            ops.rowsum(x, ws, ws_ld=ws_ld)

    def _wg_flush(self):
        if not any(self._wg_jobs.values()) and not self._rs_jobs and not self._gn_jobs and not self._pk_jobs:
            return
        if self.wgrad_stream:
            # Weight gradients are off the critical path of the backward pass: run the grouped launches on a SIDE stream so that they
            # fill the tails of (and interleave with) the input-gradient / GroupNorm / 1x1 kernels of the layers still to come.
            # Operands are kept referenced until the main stream has joined the side stream (_wg_join): the caching allocator
            # orders reuse on the allocating stream only.
            if self._wg_side is None:
                self._wg_side = torch.cuda.Stream(device=self._dev)
            main = torch.cuda.current_stream(self._dev)
            self._wg_side.wait_stream(main)
            with torch.cuda.stream(self._wg_side):
                for t, out in self._pk_jobs:                   # dY operands whose producer wrote f32 only -> pre-split images
                    ops.presplit_pack(t, out=out)
                for norm, x, y in self._gn_jobs:               # operands of the queued weight gradients (folded-GroupNorm forward)
                    norm._fwd_now(x, y)
                for x, ws, ld in self._rs_jobs:
                    ops.rowsum(x, ws, ws_ld=ld)
                for cls, jobs in self._wg_jobs.items():
                    if jobs:
                        ops.conv_wgrad_group([j[0] for j in jobs], self._dev)
            self._wg_keep.append((self._wg_jobs, self._rs_jobs, self._gn_jobs, self._pk_jobs))
            self._rs_jobs, self._gn_jobs, self._pk_jobs = [], [], []
        else:
            for t, out in self._pk_jobs:
                ops.presplit_pack(t, out=out)
            for norm, x, y in self._gn_jobs:
                norm._fwd_now(x, y)
            self._gn_jobs, self._pk_jobs = [], []
            for cls, jobs in self._wg_jobs.items():
                if jobs:
                    ops.conv_wgrad_group([j[0] for j in jobs], self._dev)
        self._wg_jobs = {}

    def _wg_join(self):
        """The main stream waits for the side-stream weight gradients (before anything reads the flat gradient)."""
        if self._wg_keep:
            torch.cuda.current_stream(self._dev).wait_stream(self._wg_side)
            self._wg_keep = []

    def _bucket_boundary(self, i: int, hook):
        """Gradient bucket `i` is complete once the weight gradients queued so far have run: hand it to the all-reduce WITHOUT stalling the
        input-gradient chain.  Round 4 joined the side stream into the main stream here (`_cs_flush`), so the backward chain of the next
        bucket waited for every weight gradient of this one -- a schedule the single-process step never had, and one nobody had timed
        (round-4 review).  Now the bucket's segmented column sums and the hook (torch.distributed's all-reduce, which orders its
        communication stream after the CURRENT stream) are issued on the weight-gradient side stream: the collective waits for exactly the
        kernels that produce the bucket, the main stream never waits.  Operands stay referenced until the join at the end of the pass.
        `VILLAN_BUCKET_JOIN=1` (or no side stream) restores the joining form."""
        if not self.wgrad_stream or os.environ.get("VILLAN_BUCKET_JOIN", "0") == "1":
            self._cs_flush()
            hook(i)
            return
        if self._wg_side is None:
            self._wg_side = torch.cuda.Stream(device=self._dev)
        self._wg_flush()
        self._wg_side.wait_stream(torch.cuda.current_stream(self._dev))      # the row-sum partials the column sums read were written on main
        with torch.cuda.stream(self._wg_side):
            self._cs_launch()
            hook(i)
        self._wg_keep.append(("bucket", i))                                  # the end-of-pass join must happen even with no job queued

    def _cs_flush(self):
        self._wg_flush()
        self._wg_join()
        self._cs_launch()

    def _cs_launch(self):
        jobs = self._cs_jobs
        if not jobs:
            return
        key = tuple(jobs)
        tab = self._cs_tables.get(key)
        if tab is None:                                          # same job list every step: uploaded once
            tab = ops.upload_table(torch.tensor(jobs, dtype=torch.int64), self._dev)
            self._cs_tables[key] = tab
        if ops._CAPTURE_TABLES is not None:
            ops._CAPTURE_TABLES.append(tab)
        ops.colsum_segmented(tab, len(jobs), self._cs_B)
        self._cs_jobs = []

    def _d_temb_buffer(self, B):
        """[B, temb_total] gradient rows of the fused time_emb_proj GEMM (persistent: its address is part of the colsum job table)."""
        t = self._d_temb.get(B)
        if t is None:
            t = torch.zeros((B, self.temb_total), device=self._dev, dtype=torch.float32)
            self._d_temb[B] = t
        else:
            ops.scale_(t, 0.0)
        return t

    def wt_view(self, prefix, M, Cc, T, fresh=True):
        """Transposed weights [C, M*T] of `prefix` for the dgrad GEMM, transposed on first use in each backward pass
        (fresh=False: only the shape is needed -- the split-precision kernel reads its own packed operand)."""
        off = self._wt_offs[prefix]
        if fresh and prefix not in self._wt_fresh:
            ops.weight_transpose(self.P[prefix + ".weight"], self._wt_buf[off:off + M * Cc * T], M, Cc, T)
            self._wt_fresh.add(prefix)
        return self._wt_buf[off:off + M * Cc * T].view(Cc, M * T)

    def _prepare_backward(self, B):
        # a backward pass that raised midway leaves queued weight-gradient / row-sum jobs behind: they must never run in THIS pass
        if self._wg_keep:
            self._wg_join()
        self._wg_jobs, self._rs_jobs, self._gn_jobs, self._pk_jobs = {}, [], [], []
        if self._wt_buf is None:
            self._wt_buf = torch.empty(self._wt_total, device=self._dev, dtype=torch.float32)
        self._wt_fresh = set()
        if self.wgrad_ws is None or getattr(self, "_ws_B", None) != B:
            need = 0
            S = self.sample_size
            for name, shape, _ in self._layout:
                if name.endswith(".weight") and len(shape) == 4:
                    M, Cc, T = shape[0], shape[1], shape[2] * shape[3]
                    for hw in {(S >> k) ** 2 for k in range(len(self.config.block_out_channels))}:
                        for md in ((B_PLAIN,) if T == 1 else (B_CONV3, B_CONV3_UP, B_CONV3_S2)):
                            need = max(need, ops.wgrad_ws_floats(M, Cc, T, B, hw, mode=md))
                        side = int(round(math.sqrt(hw)))
                        for md in (B_CONV3, B_CONV3_UP, B_CONV3_S2):
                            if T == 9 and ops.wgrad_bx3_eligible(M, Cc, side, side, md):
                                need = max(need, ops.wgrad_ws_floats(M, Cc, T, B, hw, mode=md, math_mode=1))
                        if T == 1 and ops.wgrad_bx3_eligible(M, Cc, side, side, B_PLAIN):
                            need = max(need, ops.wgrad_ws_floats(M, Cc, T, B, hw, mode=B_PLAIN, math_mode=1))
            for prefix, ch in self._qkv:
                for hw in {(S >> k) ** 2 for k in range(len(self.config.block_out_channels))}:
                    need = max(need, ops.wgrad_ws_floats(3 * ch, ch, 1, B, hw), ops.wgrad_ws_floats(ch, ch, 1, B, hw))
                    if hw % 8 == 0 and ch >= 64:
                        need = max(need, ops.wgrad_ws_floats(3 * ch, ch, 1, B, hw, mode=B_PLAIN, math_mode=1),
                                   ops.wgrad_ws_floats(ch, ch, 1, B, hw, mode=B_PLAIN, math_mode=1))
            self.wgrad_ws = torch.empty(max(need, 4), device=self._dev, dtype=torch.float32)
            self._ws_B = B

    # ------------------------------------------------------------------------------------------ forward / backward
    def forward(self, sample: torch.Tensor, timestep, return_dict: bool = False):
        x = sample
        if x.device != self._dev:
            x = x.to(self._dev)
        x = x.contiguous().float()
        B = x.shape[0]
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], device=self._dev)
        t = t.to(self._dev)
        if t.dim() == 0:
            t = t[None]
        t = t.to(torch.float32).expand(B).contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            out = _UNetFn.apply(self, x, t, self._anchor)
        else:
            out, _ = self._run_forward(x, t, save=False)
        if return_dict:
            return SimpleNamespace(sample=out)
        return (out,)

    def _new(self, B, Cc, S):
        return torch.empty((B, Cc, S, S), device=self._dev, dtype=torch.float32)

    def _run_forward(self, x, t, save):
        dev = self._dev
        B, _, S, _ = x.shape
        st = SimpleNamespace(saved=[], B=B)
        sv = st.saved
        st.marks = {}                                   # tape length at the end of the down / mid stages
        # ---- time embedding (K3) ----
        emb_sin = torch.empty((B, self.time_dim0), device=dev, dtype=torch.float32)
        ops.timestep_embedding(t, self.freqs, emb_sin, self.config.flip_sin_to_cos)
        e1 = torch.empty((B, self.temb_dim), device=dev, dtype=torch.float32)
        ops.linear(emb_sin, self.P["time_embedding.linear_1.weight"], self.P["time_embedding.linear_1.bias"], e1)
        e1a = ops.silu_fwd(e1, torch.empty_like(e1))
        emb = torch.empty_like(e1)
        ops.linear(e1a, self.P["time_embedding.linear_2.weight"], self.P["time_embedding.linear_2.bias"], emb)
        emb_act = ops.silu_fwd(emb, torch.empty_like(emb))
        st.temb_all = torch.empty((B, self.temb_total), device=dev, dtype=torch.float32)
        ops.linear(emb_act, self.Wt_all, self.bt_all, st.temb_all)
        if save:
            st.temb_saved = (emb_sin, e1, e1a, emb, emb_act)

        # ---- plan the zero-copy concat buffers: skip k is consumed by the k-th up resnet from the end ----
        up_slots = []                                   # (h_ch, skip_ch, spatial) in consumption order
        sp = S >> (len(self.down) - 1)
        for bi, blk in enumerate(self.up):
            for j in range(len(blk["res"])):
                up_slots.append((blk["h_ch"][j], blk["skip_ch"][j], sp))
            if blk["us"] is not None:
                sp *= 2
        n_skip = len(up_slots)
        cats: List[Optional[torch.Tensor]] = [None] * n_skip

        def skip_out(k, ch, spatial):
            """Output view for the k-th produced skip (consumed by up slot n_skip-1-k)."""
            h_ch, s_ch, s_sp = up_slots[n_skip - 1 - k]
            assert s_ch == ch and s_sp == spatial, (k, ch, spatial, up_slots[n_skip - 1 - k])
            buf = self._new(B, h_ch + s_ch, spatial)
            cats[n_skip - 1 - k] = buf
            return buf[:, h_ch:]

        # ---- down path ----
        k = 0
        h = skip_out(k, self._conv_in.cout, S); k += 1
        self._conv_in.fwd(x, h)
        if save:
            sv.append(("conv_in", x))
        sp = S
        for blk in self.down:
            for j, res in enumerate(blk["res"]):
                if blk["attn"]:
                    tmp = self._new(B, res.cout, sp)
                    s = res.fwd(h, tmp, st, save)
                    out = skip_out(k, res.cout, sp); k += 1
                    s2 = blk["attn"][j].fwd(tmp, out, st, save)
                    if save:
                        sv.append(("res", res, s)); sv.append(("attn", blk["attn"][j], s2))
                else:
                    out = skip_out(k, res.cout, sp); k += 1
                    s = res.fwd(h, out, st, save)
                    if save:
                        sv.append(("res", res, s))
                h = out
            if blk["ds"] is not None:
                sp //= 2
                out = skip_out(k, blk["ds"].cout, sp); k += 1
                blk["ds"].fwd(h, out)
                if save:
                    sv.append(("ds", blk["ds"], h))
                h = out
        assert k == n_skip
        st.marks["down_end"] = len(sv)
        # ---- mid ----
        tmp = self._new(B, self.mid_res[0].cout, sp)
        s = self.mid_res[0].fwd(h, tmp, st, save)
        tmp2 = self._new(B, self.mid_res[0].cout, sp)
        s2 = self.mid_attn.fwd(tmp, tmp2, st, save)
        slot = 0
        out = cats[slot][:, :up_slots[slot][0]]
        s3 = self.mid_res[1].fwd(tmp2, out, st, save)
        if save:
            sv.append(("res", self.mid_res[0], s)); sv.append(("attn", self.mid_attn, s2)); sv.append(("res", self.mid_res[1], s3))
        st.marks["mid_end"] = len(sv)
        # ---- up path ----
        final = None
        for bi, blk in enumerate(self.up):
            nres = len(blk["res"])
            for j, res in enumerate(blk["res"]):
                xin = cats[slot]
                last_of_net = (slot == n_skip - 1)
                nxt_in_block = j + 1 < nres
                # where does this resnet (or its attention) write?
                if nxt_in_block:
                    dest = cats[slot + 1][:, :up_slots[slot + 1][0]]
                elif blk["us"] is not None:
                    dest = self._new(B, res.cout, sp)
                else:
                    dest = self._new(B, res.cout, sp)
                if blk["attn"]:
                    tmp = self._new(B, res.cout, sp)
                    s = res.fwd(xin, tmp, st, save)
                    s2 = blk["attn"][j].fwd(tmp, dest, st, save)
                    if save:
                        sv.append(("res", res, s)); sv.append(("attn", blk["attn"][j], s2))
                else:
                    s = res.fwd(xin, dest, st, save)
                    if save:
                        sv.append(("res", res, s))
                h = dest
                slot += 1
            if blk["us"] is not None:
                sp *= 2
                dest = cats[slot][:, :up_slots[slot][0]]
                hp = blk["us"].us_input(h, save)
                blk["us"].fwd(hp, dest)
                if save:
                    sv.append(("us", blk["us"], hp))
                h = dest
            else:
                final = h
        # ---- out ----
        a = self._new(B, self.norm_out.ch, S)
        mo, ro = self.norm_out.fwd(final, a)
        out = self._new(B, self.out_channels, S)
        self._conv_out.fwd(a, out)
        if save:
            sv.append(("out", final, a, mo, ro))
            st.up_slots, st.cats = up_slots, cats
        return out, st

    def _run_backward(self, st, dout):
        dev = self._dev
        B = st.B
        self._prepare_backward(B)
        self._cs_begin(B)
        st.d_temb_all = self._d_temb_buffer(B)
        up_slots = st.up_slots
        n_skip = len(up_slots)
        dcats: List[Optional[torch.Tensor]] = [None] * n_skip     # gradient wrt each concat buffer
        sv = st.saved
        # ---- out ----
        _, final, a, mo, ro = sv.pop()
        da = torch.empty_like(a)
        self._conv_out.bwd(dout, a, da)
        g = torch.empty(final.shape, device=dev, dtype=torch.float32)
        fuse = self.fuse_gn_bwd
        # g_rs: per-image channel sums of g ([B, C] view) when the kernel that wrote g also summed it (vd_groupnorm_bwd_fused), else None
        g_rs = self.scratch_bc(B, final.shape[1]).view(B, final.shape[1]) if fuse else None

        def wants_ps(norm, shape):
            """Is the next record on the tape (the consumer of the gradient about to be produced) a pre-split ResnetBlock, and can `norm`'s backward
            kernel write its dx as a pre-split image too?  Then the gradient is handed on in both forms (g, g_ps)."""
            if not sv or not norm.ps_ok(shape[2] * shape[3]):
                return None
            nxt = sv[-1]
            if nxt[0] == "res" and isinstance(nxt[2][5], ops.PreSplit):
                return ops.presplit_empty(shape, dev)
            if (nxt[0] == "us" and self.presplit and self.conv_math == "bf16x3" and self.group_wgrad
                    and ops.wgrad_presplit_ok(shape[0], nxt[1].cin, nxt[1].cout, shape[3], B_CONV3_UP)):
                return ops.presplit_empty(shape, dev)            # (the upsampler's weight gradient reads it; its input gradient reads f32)
            return None

        # g_ps: g as a pre-split image when its producer wrote one (consumed by a pre-split block's conv2 gradients), else None
        g_ps = wants_ps(self.norm_out, final.shape)
        self.norm_out.bwd(da, final, mo, ro, g, rowsum=g_rs, dx_ps=g_ps)
        # `g` is the gradient wrt the output of the most recent forward op; walk the tape backwards.
        slot = n_skip
        hook = self.bucket_ready_hook
        while sv:
            if hook is not None:                                  # gradient buckets complete in the order up|out, mid, down
                if len(sv) == st.marks["mid_end"]:
                    self._bucket_boundary(0, hook)
                elif len(sv) == st.marks["down_end"]:
                    self._bucket_boundary(1, hook)
            rec = sv.pop()
            kind = rec[0]
            if kind in ("res", "attn"):
                layer, saved = rec[1], rec[2]
                x = saved[0]
                is_cat_input = kind == "res" and any(x.data_ptr() == c.data_ptr() and x.shape == c.shape for c in st.cats)
                dx = torch.empty(x.shape, device=dev, dtype=torch.float32)
                rs = self.scratch_bc(B, x.shape[1]).view(B, x.shape[1]) if fuse else None
                dxp = wants_ps(layer.norm1 if kind == "res" else layer.norm, x.shape) if fuse else None
                if is_cat_input:
                    slot -= 1
                    dcats[slot] = dx
                    layer.bwd(saved, g, dx, st, dout_rs=g_rs, dx_rs=rs, dout_ps=g_ps, dx_ps=dxp)
                    g = dx[:, :up_slots[slot][0]]
                    g_rs = rs[:, :up_slots[slot][0]] if fuse else None
                    g_ps = dxp.channels(0, up_slots[slot][0]) if dxp is not None and up_slots[slot][0] % 8 == 0 else None
                elif fuse:                                        # the skip connection's gradient rides in the block's last GroupNorm backward
                    layer.bwd(saved, g, dx, st, dout_rs=g_rs, extra2=self._skip_grad(x, st, dcats), dx_rs=rs, dout_ps=g_ps, dx_ps=dxp)
                    g, g_rs, g_ps = dx, rs, dxp
                else:
                    layer.bwd(saved, g, dx, st, dout_ps=g_ps)
                    g, g_ps = dx, None
                    g = self._add_skip_grad(g, x, st, dcats)
            elif kind == "us":
                layer, x = rec[1], rec[2]
                dx = torch.empty(x.shape, device=dev, dtype=torch.float32)
                layer.bwd(g, x, dx, bias_ws=g_rs, dout_ps=g_ps)
                g, g_rs, g_ps = dx, None, None
            elif kind == "ds":
                layer, x = rec[1], rec[2]
                dx = torch.empty(x.shape, device=dev, dtype=torch.float32)
                layer.bwd(g, x, dx, bias_ws=g_rs)
                g, g_rs, g_ps = self._add_skip_grad(dx, x, st, dcats), None, None
            elif kind == "conv_in":
                self._conv_in.bwd(g, rec[1], None, bias_ws=g_rs)
            else:
                raise RuntimeError(kind)
        self._cs_flush()                                          # d_temb_all's partials may sit on the side stream
        if hook is not None:
            hook(2)
        # ---- time embedding backward ----
        emb_sin, e1, e1a, emb, emb_act = st.temb_saved
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
        ops.linear_wgrad(d_e1, emb_sin, self.G["time_embedding.linear_1.weight"], accumulate=True)
        ops.colsum(d_e1, self.G["time_embedding.linear_1.bias"], B, self.temb_dim, accumulate=True)
        self._cs_flush()
        if hook is not None:
            hook(3)

    def _skip_grad(self, x, st, dcats):
        """If `x` is a skip tensor living in a concat buffer: the gradient that flowed into it through the up path (a channel
        slice of that buffer's gradient), else None."""
        for k, c in enumerate(st.cats):
            h_ch = st.up_slots[k][0]
            if x.shape[1:] == c[:, h_ch:].shape[1:] and x.data_ptr() == c[:, h_ch:].data_ptr():
                return dcats[k][:, h_ch:]
        return None

    def _add_skip_grad(self, g, x, st, dcats):
        """g += the up path's gradient of the skip tensor `x` (whose gradient `g` was just produced), if it is one."""
        sg = self._skip_grad(x, st, dcats)
        if sg is not None:
            ops.add_strided(g, sg, accumulate=True)
        return g
