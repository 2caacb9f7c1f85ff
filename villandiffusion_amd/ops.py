"""Tensor-level wrappers over the C ABI (include/villan_hip.h).

torch is plumbing here: it owns device memory and the current HIP stream; every computation below is one of
the hand-written gfx950 kernels.  Image tensors are [B, C, H, W] views whose (C, H, W) part is contiguous and
whose batch stride may be larger than C*H*W (channel slices of a concat buffer).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch

from . import lib as L
from .lib import (A_COL, A_ROW, B_CONV3, B_CONV3_DIL, B_CONVG, B_CONV3_S2, B_CONV3_T, B_CONV3_UP, B_KCONTIG, B_PLAIN,
                  GemmDesc, WgradDesc)


def _lib():
    L.require_device()
    return L.load()


# ---- optional per-launch timing of the MFMA kernels (bench.py roofline leg): HIP events on the launch stream ----
_PROF = None          # list of (kernel_symbol, algorithmic_flops, start_event, end_event) while enabled

_B_NAMES = {0: "PLAIN", 1: "KCONTIG", 2: "CONV3", 3: "CONV3_T", 4: "CONV3_S2", 5: "CONV3_UP", 6: "CONV3_DIL", 7: "CONVG"}
_TILE_NAMES = {1: "128x128", 2: "64x128", 3: "64x64", 4: "patch128x128"}


def profile_start():
    global _PROF
    _PROF = []


def profile_stop():
    global _PROF
    rec, _PROF = _PROF, None
    return rec


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _timed(name, amount, kind, call, nbytes=None):
    """Run `call()`; while profiling is on, bracket it with HIP events on the launch stream and record a dict
    {name, flops, bytes, e0, e1, kind}: kind "mfma" -> amount = algorithmic FLOPs (nbytes = algorithmic bytes, optional),
    kind "hbm" -> amount = algorithmic bytes (every operand read once / written once)."""
    if _PROF is None:
        return call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = call()
    e1.record()
    _PROF.append({"name": name, "flops": float(amount) if kind == "mfma" else 0.0, "bytes": float(nbytes if nbytes is not None else (amount if kind == "hbm" else 0.0)),
                  "e0": e0, "e1": e1, "kind": kind})
    return r


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _img(t: torch.Tensor):
    """(B, C, H, W, bstride) of an image view; inner CHW must be contiguous."""
    assert t.dtype == torch.float32 and t.dim() == 4, (t.dtype, t.shape)
    B, Cc, H, W = t.shape
    st = t.stride()
    assert (st[1], st[2], st[3]) == (H * W, W, 1) or Cc * H * W == 0, f"inner CHW not contiguous: {t.shape} {st}"
    return B, Cc, H, W, (st[0] if B > 1 else Cc * H * W)


# --------------------------------------------------------------------------------------------- pre-split activation images
class PreSplit:
    """A [B, C, H, W] activation held as its PRE-SPLIT image (include/villan_hip.h "PRE-SPLIT activation images": per pixel and channel octet the
    bf16 hi / lo halves the split-precision kernels contract, written once by the producer).  `t` is the float32 tensor whose storage carries the
    image: same shape, bytes, batch stride and channel-octet offsets as the f32 tensor it replaces -- only the meaning of the bytes differs, so
    nothing but a kernel that declares a pre-split operand may read it."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        assert t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] % 8 == 0, (t.dtype, t.shape)
        self.t = t

    shape = property(lambda self: self.t.shape)
    device = property(lambda self: self.t.device)

    def stride(self, *a):
        return self.t.stride(*a)

    def data_ptr(self):
        return self.t.data_ptr()

    def channels(self, lo: int, hi: int) -> "PreSplit":
        """The channel slice [lo, hi) (whole octets): an O8 image of a channel range is the same bytes as the f32 slice."""
        assert lo % 8 == 0 and hi % 8 == 0
        return PreSplit(self.t[:, lo:hi])


def _unwrap(x):
    """(tensor, is_presplit)"""
    return (x.t, True) if isinstance(x, PreSplit) else (x, False)


def presplit_empty(shape, device) -> PreSplit:
    return PreSplit(torch.empty(shape, device=device, dtype=torch.float32))


def presplit_pack(x: torch.Tensor, out: Optional[PreSplit] = None) -> PreSplit:
    Bn, Cc, H, W, xbs = _img(x)
    out = presplit_empty(x.shape, x.device) if out is None else out
    assert tuple(out.shape) == tuple(x.shape)
    _timed("presplit_pack (presplit_pack_kernel)", 8.0 * x.numel(), "hbm", lambda: L.check(
        _lib().vd_presplit_pack(_p(x), _p(out.t), Bn, Cc, H * W, xbs, _img(out.t)[4], _s()), "vd_presplit_pack"))
    return out


def presplit_unpack(y: PreSplit, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    Bn, Cc, H, W, ybs = _img(y.t)
    out = torch.empty(y.shape, device=y.device, dtype=torch.float32) if out is None else out
    L.check(_lib().vd_presplit_unpack(_p(y.t), _p(out), Bn, Cc, H * W, ybs, _img(out)[4], _s()), "vd_presplit_unpack")
    return out


def groupnorm_presplit_ok(Cc: int, HW: int, G: int) -> bool:
    return bool(_lib().vd_groupnorm_fwd_presplit_ok(Cc, HW, G))


def groupnorm_fwd_presplit(x, gamma, beta, y: PreSplit, mean, rstd, G, eps, silu) -> PreSplit:
    """y = the pre-split image of silu?(GroupNorm(x)); mean / rstd as groupnorm_fwd."""
    Bn, Cc, H, W, xbs = _img(x)
    assert tuple(y.shape) == tuple(x.shape) and mean.numel() >= Bn * G and rstd.numel() >= Bn * G
    _timed("groupnorm_fwd_presplit (gn_fwd_ps_kernel<*>)", 8.0 * x.numel(), "hbm", lambda: L.check(
        _lib().vd_groupnorm_fwd_presplit(_p(x), _p(gamma), _p(beta), _p(y.t), _p(mean), _p(rstd), Bn, Cc, H * W, G, eps, int(silu), xbs,
                                         _img(y.t)[4], _s()), "vd_groupnorm_fwd_presplit"))
    return y


_PS_CONV_OK = {}


def conv_presplit_ok(Bn: int, Cin: int, Cout: int, OH: int, OW: int, mode: int) -> bool:
    """Would conv3x3(PreSplit input [Bn, Cin, ..] -> [Bn, Cout, OH, OW], mode, a_packed=...) be taken by the persistent 16x16x32 kernel, the only
    reader of pre-split images?  (vd_gemm_tile() == 18 with b_presplit = 1; asked once per shape.)"""
    key = (Bn, Cin, Cout, OH, OW, mode)
    ok = _PS_CONV_OK.get(key)
    if ok is None:
        H, W = (OH // 2, OW // 2) if mode == B_CONV3_UP else (OH, OW)
        d = GemmDesc()
        d.A = d.B = d.D = d.a_packed = 64                  # (non-null placeholders: vd_gemm_tile() only looks at shapes, strides and alignment)
        d.a_packed_mpad = (Cout + 127) // 128 * 128
        d.M, d.N, d.K = Cout, Bn * OH * OW, Cin * 9
        d.a_mode, d.b_mode, d.NP = A_ROW, mode, OH * OW
        d.C, d.H, d.W, d.OH, d.OW = Cin, H, W, OH, OW
        d.alpha = 1.0
        d.lda, d.b_bstride, d.ldd, d.d_bstride = Cin * 9, Cin * H * W, OH * OW, Cout * OH * OW
        d.b_presplit = 1
        ok = _PS_CONV_OK[key] = int(_lib().vd_gemm_tile(C.byref(d))) == 18
    return ok


def wgrad_presplit_ok(Bn: int, Cin: int, Cout: int, S: int, mode: int = B_CONV3) -> bool:
    """Is there a grouped pre-split weight-gradient kernel for a 3x3 convolution Cin -> Cout with S x S OUTPUTS (mode B_CONV3 or B_CONV3_UP)?"""
    d = WgradDesc()
    d.dY = d.X = d.dW = 64
    d.M, d.C, d.T, d.nb, d.NP = Cout, Cin, 9, Bn, S * S
    d.OH = d.OW = S
    d.H = d.W = S // 2 if mode == B_CONV3_UP else S
    d.mode, d.accumulate, d.math, d.presplit = mode, 1, 1, 3
    d.dy_bstride, d.x_bstride = Cout * S * S, Cin * d.H * d.W
    return int(_lib().vd_conv_wgrad_group_class(C.byref(d))) >= 3000


def groupnorm_bwd_presplit(dy, x, mean, rstd, gamma, beta, dx, dx_ps, dgamma_ws, dbeta_ws, G, silu, extra=None, extra2=None, rowsum=None,
                           rowsum_ld=None):
    """groupnorm_bwd whose dx goes out as f32 (`dx`, may be None), as a pre-split image (`dx_ps`: PreSplit, may be None) or both."""
    Bn, Cc, H, W, xbs = _img(x)
    dbs = _img(dy)[4]
    assert dx is not None or dx_ps is not None
    dxbs = _img(dx)[4] if dx is not None else 0
    psbs = _img(dx_ps.t)[4] if dx_ps is not None else 0
    assert dx is None or dx.shape == x.shape
    assert dx_ps is None or tuple(dx_ps.shape) == tuple(x.shape)
    ebs = _img(extra)[4] if extra is not None else 0
    e2bs = _img(extra2)[4] if extra2 is not None else 0
    assert dgamma_ws.numel() >= Bn * Cc and dbeta_ws.numel() >= Bn * Cc
    assert (extra is None or extra.shape == x.shape) and (extra2 is None or extra2.shape == x.shape)
    rld = 0
    if rowsum is not None:
        rld = Cc if rowsum_ld is None else rowsum_ld
        assert rld >= Cc
    nbytes = (8.0 + (4.0 if dx is not None else 0.0) + (4.0 if dx_ps is not None else 0.0) + (4.0 if extra is not None else 0.0)
              + (4.0 if extra2 is not None else 0.0)) * x.numel()
    _timed("groupnorm_bwd_presplit (gn_bwd_ps_kernel<*>)", nbytes, "hbm", lambda: L.check(
        _lib().vd_groupnorm_bwd_presplit(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(extra), _p(extra2), _p(dx),
                                         _p(dx_ps.t) if dx_ps is not None else None, _p(dgamma_ws), _p(dbeta_ws), _p(rowsum), Bn, Cc, H * W, G,
                                         int(silu), dbs, xbs, ebs, e2bs, dxbs, psbs, rld, _s()), "vd_groupnorm_bwd_presplit"))
    return dx_ps if dx_ps is not None else dx


# --------------------------------------------------------------------------------------------- GEMM family
_GEMM_WS = {}
_WS_SLOT = 0
AUX_WS_SLOT = -1       # the UNet's auxiliary stream (unet.aux_scope): outside the non-negative slots pipelines.sample_concurrent hands to its streams


class ws_slot:
    """Launch sequences that run CONCURRENTLY on different streams (the sampler's interleaved chunks, pipelines.sample_concurrent) must not share
    the split-K workspace: each takes its own slot.  `with ops.ws_slot(k): ...` around everything that launches (or captures) on that stream."""

    def __init__(self, slot: int):
        self.slot, self.prev = int(slot), 0

    def __enter__(self):
        global _WS_SLOT
        self.prev, _WS_SLOT = _WS_SLOT, self.slot
        return self

    def __exit__(self, *exc):
        global _WS_SLOT
        _WS_SLOT = self.prev
        return False


def gemm_ws_buffer(device, slot: int = None):
    """The current split-K workspace tensor of (device, slot), or None (graph owners compare it by identity to know their capture is still valid)."""
    return _GEMM_WS.get((device, _WS_SLOT if slot is None else slot))


def _gemm_ws(n_floats: int, device):
    """Per-(device, slot) split-K workspace, grown on demand (launches on one stream are ordered, so one stream's launches share it)."""
    key = (device, _WS_SLOT)
    t = _GEMM_WS.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.empty(max(n_floats, 1 << 22), device=device, dtype=torch.float32)
        _GEMM_WS[key] = t
    return t


def gemm(A, B, D, *, M, N, K, a_mode=A_ROW, b_mode=B_PLAIN, NP=None, lda=0, a_bstride=0, ldb=0, b_bstride=0,
         ldd=0, d_bstride=0, bias=None, bias_on_n=False, rowadd=None, rowadd_bstride=0, residual=None,
         res_bstride=0, conv=None, alpha=1.0, d_trans=False, accumulate=False, tile=0, debug=0, pad=0, nb2=0, a_b2stride=0,
         b_b2stride=0, d_b2stride=0, gn_ss=None, a_packed=None, math_mode=0, pool2=False, convg=None, act=0, gn_part=None, act_out=None,
         b_presplit=False):
    """gn_part: optional [B, NP // 256, M, 2] buffer for the per-tile channel sums of the result (vd_gemm_desc.gn_part); it is filled only when the
    launch goes to the 16x16x32 split-precision convolution -- ops.GN_PART_WRITTEN tells the caller right after the call."""
    global GN_PART_WRITTEN
    d = GemmDesc()
    d.act = act
    d.b_presplit = int(b_presplit)
    if convg is not None:
        d.kh, d.kw, d.conv_stride, d.pad_h, d.pad_w = convg
    d.pad = pad
    d.pool2 = int(pool2)
    d.math = math_mode
    a_packed16 = None
    if isinstance(a_packed, tuple):            # (split-precision operand, f16 operand): opt-in mixed precision, decided per problem below
        a_packed, a_packed16 = a_packed
    if a_packed is not None:
        d.a_packed, d.a_packed_mpad = a_packed.data_ptr(), (M + 127) // 128 * 128
    d.gn_ss = _p(gn_ss)
    d.nb2, d.a_b2stride, d.b_b2stride, d.d_b2stride = nb2, a_b2stride, b_b2stride, d_b2stride
    d.A, d.B, d.D = _p(A), _p(B), _p(D)
    d.bias, d.rowadd, d.residual = _p(bias), _p(rowadd), _p(residual)
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.b_mode = a_mode, b_mode
    d.NP = N if NP is None else NP
    if conv is not None:
        d.C, d.H, d.W, d.OH, d.OW = conv
    d.bias_on_n, d.d_trans, d.accumulate, d.tile = int(bias_on_n), int(d_trans), int(accumulate), tile
    d.alpha = alpha
    d.debug = debug
    d.lda, d.a_bstride, d.ldb, d.b_bstride = lda, a_bstride, ldb, b_bstride
    d.ldd, d.d_bstride, d.res_bstride, d.rowadd_bstride = ldd, d_bstride, res_bstride, rowadd_bstride
    lib = _lib()
    need = lib.vd_gemm_ws_floats(C.byref(d))
    if need > 0:
        d.ws = _gemm_ws(need, D.device).data_ptr()
    elif FORCE_WS is not None:                 # diagnostic builds only (tools/k32p_stamps.py: the stamp buffer travels in the unused ws pointer)
        d.ws = FORCE_WS.data_ptr()
    global LAST_GEMM_TILE, LAST_GEMM_MATH
    if a_packed16 is not None and gn_part is None:
        # f16 operands are read by the persistent 16x16x32 kernels only (vd_gemm_tile 18 / 19 with math = 2); every other problem keeps the
        # split-precision operand
        d.a_packed, d.math = a_packed16.data_ptr(), 2
        if lib.vd_gemm_tile(C.byref(d)) not in (18, 19):
            d.a_packed, d.math = a_packed.data_ptr(), math_mode
    LAST_GEMM_MATH = d.math
    LAST_GEMM_TILE = lib.vd_gemm_tile(C.byref(d))            # kernel family the library picks for this problem (tests assert on it)
    global ACT_OUT_WRITTEN
    ACT_OUT_WRITTEN = False
    if act_out is not None and gn_ss is not None and a_packed16 is None:
        # side output of the persistent kernel's GroupNorm-folding loader (vd_gemm_desc.act_out): taken only where that kernel runs the problem
        d.act_out, d.act_bstride = act_out.data_ptr(), _img(act_out)[4]
        if lib.vd_gemm_tile(C.byref(d)) == 18:
            ACT_OUT_WRITTEN = True
        else:
            d.act_out, d.act_bstride = None, 0
        LAST_GEMM_TILE = lib.vd_gemm_tile(C.byref(d))
    GN_PART_WRITTEN = gn_part is not None and not pool2 and LAST_GEMM_TILE in (17, 18)
    if GN_PART_WRITTEN:
        assert gn_part.is_contiguous() and gn_part.numel() >= (N // 256) * M * 2
        d.gn_part = gn_part.data_ptr()
    if _PROF is None:
        L.check(lib.vd_gemm(C.byref(d), _s()), "vd_gemm")
        return D
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.vd_gemm(C.byref(d), _s()), "vd_gemm")
    e1.record()
    flops = 2.0 * M * N * K * (0.25 if b_mode == B_CONV3_DIL else 1.0)     # DIL: 3/4 of the taps are structural zeros
    # algorithmic bytes: B (the activation operand) read once, D written once (+ residual read), A read once
    if conv is not None:
        nb_ = N // d.NP
        b_elems = nb_ * d.C * d.H * d.W
        a_elems = M * K
    else:
        b_elems = K * N
        a_elems = M * K * (N // d.NP if a_bstride else 1)
    d_elems = M * N // (4 if pool2 else 1)
    nbytes = 4.0 * (b_elems + d_elems * (2 if residual is not None else 1) + a_elems)
    tl = LAST_GEMM_TILE
    if tl == 10:
        name = f"gemm_bx3_act_kernel<{int(a_mode == A_ROW)}, {int(b_mode == B_KCONTIG)}>"
    elif tl == 19:
        # (the 256 x 128 tile where M % 256 == 0: vd_launch_gemm1x1_k32p; symbol names as rocprofv3 prints them)
        big = d.math != 2 and M % 256 == 0 and N % 128 == 0 and os.environ.get("VD_G32P_BM256", "1") != "0"
        name = "gemm1x1_k32p_kernel<true, 128>" if d.math == 2 else f"gemm1x1_k32p_kernel<false, {256 if big else 128}>"
    elif tl in (9, 11, 13):
        name = "gemm_bx3_persist_kernel" if tl == 11 else f"gemm_bx3_kernel<{512 if tl == 13 else 256}>"
    elif tl == 17:
        md = (3 if gn_ss is not None else 0) if b_mode == B_CONV3 else (1 if b_mode == B_CONV3_T else 2)
        name = f"conv3_k32_kernel<{d.OW}, {md}>"
    elif tl == 18:          # the persistent kernel: template width 16 (16x16 images) or 32 (8-row x 32-column segments of any image); image width beside it
        md = (3 if gn_ss is not None else 0) if b_mode == B_CONV3 else (1 if b_mode == B_CONV3_T else 2)
        name = (f"conv3_k32p_kernel<{16 if d.OW == 16 else 32}, {md}, true, true, {'true' if d.math == 2 else 'false'}, {'true' if d.b_presplit else 'false'}>"
                + (f"@{d.OW}" if d.OW > 32 else ""))
    elif tl == 20:          # the whole-K kernel of the 8x8 / 4x4 levels
        imgs = 2 if d.OW == 8 else 4
        name = f"conv3_sm_kernel<{d.OW}, {1 if b_mode == B_CONV3_T else 0}, {imgs}, 1>"
    elif tl in (8, 12, 15, 16):
        md = (3 if gn_ss is not None else 0) if b_mode == B_CONV3 else (1 if b_mode == B_CONV3_T else (4 if b_mode == B_CONV3_S2 else 2))
        name = f"conv3_bx3_kernel<{d.OW if d.OW <= 64 else 128}, {md}, {4 if tl == 15 else 2}, {256 if tl == 8 else 512}, 2>"
    elif tl in (4, 6):     # symbol names as rocprofv3 prints them
        tw = d.OW if d.OW <= 64 else 128          # tile width (template W): row segments of wider images
        md = (3 if gn_ss is not None else 0) if b_mode == B_CONV3 else (1 if b_mode == B_CONV3_T else (4 if b_mode == B_CONV3_S2 else 2))
        name = f"conv3_patch_kernel<{tw}, {md}, {4 if tl == 6 else 2}>"
    elif tl == 5:
        name = f"gemm_plain_kernel<{a_mode}>"
    elif tl == 7:
        few = d.W % 4 == 0 and d.C >= 16 and d.b_bstride % 4 == 0 and d.d_bstride % 4 == 0 and d.ldd % 4 == 0      # fewout_eligible() of vd_gemm.hip
        name = "conv3_fewout_kernel<4, 8>" if few else f"conv3_smallm_kernel<{32 if d.W % 32 == 0 else 16}, 4>"
    else:
        name = f"gemm_kernel<{_TILE_NAMES[tl]},{'ROW' if a_mode == A_ROW else 'COL'},{_B_NAMES[b_mode]}>"
    _PROF.append({"name": name, "flops": flops, "bytes": nbytes, "e0": e0, "e1": e1, "kind": "mfma", "shape": (M, K, d.NP, N // d.NP, d.OH, d.OW)})
    return D


_CONV_OUT = {B_CONV3: lambda h, w: (h, w), B_CONV3_T: lambda h, w: (h, w), B_CONV3_S2: lambda h, w: (h // 2, w // 2),
             B_CONV3_UP: lambda h, w: (2 * h, 2 * w), B_CONV3_DIL: lambda h, w: (2 * h, 2 * w)}


def conv3_pack_weights(w2d, M, Cc, transposed=False, out=None, taps=9):
    """Split-precision operand of conv3x3 / conv1x1 (a_packed=...): the [M, Cc*taps] weights as bf16 (hi, lo) pairs in MFMA
    fragment order.  transposed=True packs the dgrad operand A'[c_in][m_out] straight from the forward weights w2d
    [m_out, c_in*taps] (then M = c_in, Cc = m_out)."""
    lib = _lib()
    nbytes = lib.vd_conv3_packed_bytes(M, Cc, taps)
    assert nbytes > 0, (M, Cc, taps)
    if out is None:
        out = torch.empty(nbytes // 4, device=w2d.device, dtype=torch.int32)
    assert out.numel() * out.element_size() >= nbytes and w2d.is_contiguous()
    rs, cs = (taps, M * taps) if transposed else (Cc * taps, taps)
    L.check(lib.vd_conv3_pack_weights(_p(w2d), _p(out), M, Cc, taps, rs, cs, _s()), "vd_conv3_pack_weights")
    return out


def conv3_pack_weights_f16(w2d, M, Cc, transposed=False, taps=9):
    """f16 operand (vd_gemm_desc.math = 2) of ONE convolution -- tests and tools; the network packs all of them in one launch
    (unet._PackedConvWeights(f16=True)).  Pass it to conv3x3 / conv1x1 / gemm as a_packed=(split-precision operand, this)."""
    nbytes = _lib().vd_conv3_packed_bytes(M, Cc, taps) // 2
    assert nbytes > 0 and w2d.is_contiguous(), (M, Cc, taps)
    out = torch.empty(nbytes // 4, device=w2d.device, dtype=torch.int32)
    rs, cs = (taps, M * taps) if transposed else (Cc * taps, taps)
    mpad = (M + 127) // 128 * 128
    tab = torch.tensor([[w2d.data_ptr(), out.data_ptr(), M, Cc, rs, cs, 0, taps]], dtype=torch.int64).to(w2d.device)
    conv3_pack_weights_f16_multi(tab, 1, (mpad * (Cc // 16) * 2 + 255) // 256)
    torch.cuda.current_stream(w2d.device).synchronize()       # the one-row table is a temporary
    return out


def conv3_pack_weights_f16_multi(table, n_jobs, total_blocks):
    """The same job table, f16 operands (vd_gemm_desc.math = 2): half the bytes per job."""
    assert table.dtype == torch.int64 and table.is_contiguous() and table.numel() >= 8 * n_jobs
    L.check(_lib().vd_conv3_pack_weights_f16_multi(_p(table), n_jobs, total_blocks, _s()), "vd_conv3_pack_weights_f16_multi")


def conv3_pack_weights_multi(table, n_jobs, total_blocks):
    """table: device int64 [n_jobs, 8] = (W address, packed address, M, C, row_stride, chan_stride, first workgroup, 0)."""
    assert table.dtype == torch.int64 and table.is_contiguous() and table.numel() >= 8 * n_jobs
    L.check(_lib().vd_conv3_pack_weights_multi(_p(table), n_jobs, total_blocks, _s()), "vd_conv3_pack_weights_multi")


# Bumped by every raw-pointer parameter update (vd_adam_step writes behind torch's version counters): caches derived from the
# weights (packed split-precision operands) key on (flat_param._version, WEIGHTS_EPOCH).
WEIGHTS_EPOCH = 0


def bx3_eligible(M, Cc, OH, OW, mode) -> bool:
    """Problems the split-precision convolution kernel takes (vd_gemm_desc.a_packed)."""
    if mode == B_CONV3_S2:                                  # stride 2 (OH, OW: the OUTPUT, half the input): square 4x4 .. 32x32 outputs, 64 / 128 k wide ones
        return Cc % 16 == 0 and M >= 64 and OH == OW and (OW in (4, 8, 16, 32, 64) or (OW >= 128 and OW % 128 == 0)) \
            and os.environ.get("VD_BX3_S2_OFF", "0") in ("", "0")
    if mode not in (B_CONV3, B_CONV3_T, B_CONV3_UP) or Cc % 16 != 0 or M < 64 or (OW == 4 and mode == B_CONV3_UP):
        return False
    if OW == 64 or (OW >= 128 and OW % 128 == 0):           # row-segment tiles of wide images
        return (OH * OW) % 128 == 0
    return OH == OW and OW in (4, 8, 16, 32)


def bx3_pool2_eligible(M, Cc, OH, OW, nb) -> bool:
    """The split-precision stride-1 dgrad can add the 2x2 blocks of its output in the epilogue (vd_gemm_desc.pool2): 16x16 / 32x32 outputs
    on an unsplit grid (>= 256 tiles of 128 channels x 128 pixels; below that the kernel splits the channel loop over workgroups)."""
    return bx3_eligible(M, Cc, OH, OW, B_CONV3_T) and OH == OW and OW in (16, 32) and ((M + 127) // 128) * ((nb * OH * OW + 127) // 128) >= 256


GN_PART_WRITTEN = False
ACT_OUT_WRITTEN = False
LAST_GEMM_TILE = 0
LAST_GEMM_MATH = 0
FORCE_WS = None


def conv3x3(x, w2d, bias, out, mode=B_CONV3, rowadd=None, rowadd_bstride=0, residual=None, accumulate=False, tile=0, debug=0,
            pad=0, gn_ss=None, a_packed=None, pool2=False, gn_part=None, act_out=None):
    """out[b] = W (*) gather_mode(x[b]) + bias (+ rowadd[b,:,None,None]) (+ residual).  w2d: [M, C*9].
    pad: stride-2 mode only (0: zero pad (0,1,0,1); 1: symmetric padding 1).
    pool2 (B_CONV3_T with a_packed only): `out` has HALF the resolution and receives the 2x2 block sums of the result."""
    x, ps = _unwrap(x)                          # a PreSplit input: the persistent 16x16x32 kernel copies its (hi, lo) units (vd_gemm_desc.b_presplit)
    Bn, Cc, H, W, xbs = _img(x)
    M = w2d.shape[0]
    assert w2d.shape[1] == Cc * 9 and w2d.is_contiguous()
    OH, OW = _CONV_OUT[mode](H, W)
    Bo, Mo, OHo, OWo, obs = _img(out)
    if pool2:
        assert (Bo, Mo, OHo, OWo) == (Bn, M, OH // 2, OW // 2) and mode == B_CONV3_T and a_packed is not None, (out.shape, (Bn, M, OH, OW))
        assert bias is None and rowadd is None and residual is None and not accumulate
    else:
        assert (Bo, Mo, OHo, OWo) == (Bn, M, OH, OW), (out.shape, (Bn, M, OH, OW))
    rbs = 0
    if residual is not None:
        rbs = _img(residual)[4]
        assert residual.shape == out.shape
    return gemm(w2d, x, out, M=M, N=Bn * OH * OW, K=Cc * 9, b_mode=mode, NP=OH * OW, lda=Cc * 9, b_bstride=xbs,
                ldd=OHo * OWo, d_bstride=obs, bias=bias, rowadd=rowadd, rowadd_bstride=rowadd_bstride,
                residual=residual, res_bstride=rbs, conv=(Cc, H, W, OH, OW), accumulate=accumulate, tile=tile, debug=debug,
                pad=pad, gn_ss=gn_ss, a_packed=a_packed, pool2=pool2, gn_part=gn_part, act_out=act_out, b_presplit=ps)


def conv2d_general(x, w2d, bias, out, kh, kw, stride=1, pad_h=0, pad_w=0, relu=False):
    """out[b] = act(W (*) x[b] + bias) for a kh x kw convolution with zero padding (pad_h, pad_w) and stride 1 | 2 -- the InceptionV3
    convolutions of the FID measure (w2d: [M, C*kh*kw] with BatchNorm folded in).  1x1 / stride 1 goes to the plain GEMM."""
    Bn, Cc, H, W, xbs = _img(x)
    M = w2d.shape[0]
    assert w2d.shape[1] == Cc * kh * kw and w2d.is_contiguous()
    OH, OW = (H + 2 * pad_h - kh) // stride + 1, (W + 2 * pad_w - kw) // stride + 1
    Bo, Mo, OHo, OWo, obs = _img(out)
    assert (Bo, Mo, OHo, OWo) == (Bn, M, OH, OW), (out.shape, (Bn, M, OH, OW))
    if kh == 1 and kw == 1 and stride == 1 and pad_h == 0 and pad_w == 0:
        return gemm(w2d, x, out, M=M, N=Bn * OH * OW, K=Cc, b_mode=B_PLAIN, NP=OH * OW, lda=Cc, ldb=H * W, b_bstride=xbs, ldd=OH * OW,
                    d_bstride=obs, bias=bias, act=int(relu))
    return gemm(w2d, x, out, M=M, N=Bn * OH * OW, K=Cc * kh * kw, b_mode=B_CONVG, NP=OH * OW, lda=Cc * kh * kw, b_bstride=xbs, ldd=OH * OW,
                d_bstride=obs, bias=bias, conv=(Cc, H, W, OH, OW), convg=(kh, kw, stride, pad_h, pad_w), act=int(relu))


def pool3(x, out, stride=1, pad=0, mode="max"):
    """3x3 max pooling / average pooling that excludes the zero padding from the divisor (count_include_pad=False)."""
    Bn, Cc, H, W, xbs = _img(x)
    OH, OW = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    Bo, Co, OHo, OWo, obs = _img(out)
    assert (Bo, Co, OHo, OWo) == (Bn, Cc, OH, OW), (out.shape, (Bn, Cc, OH, OW))
    return _timed("pool3_kernel", 4.0 * (x.numel() + out.numel()), "hbm",
                  lambda: L.check(_lib().vd_pool3(_p(x), _p(out), Bn, Cc, H, W, stride, pad, 0 if mode == "max" else 1, xbs, obs, _s()), "vd_pool3"))


def resize_bilinear(x, out, mul=1.0, add=0.0):
    """out = mul * F.interpolate(x, out.shape[-2:], mode="bilinear", align_corners=False) + add  (x, out contiguous NCHW)."""
    assert x.is_contiguous() and out.is_contiguous() and x.shape[:2] == out.shape[:2] and x.dtype == out.dtype == torch.float32
    L.check(_lib().vd_resize_bilinear(_p(x), _p(out), x.shape[0] * x.shape[1], x.shape[2], x.shape[3], out.shape[2], out.shape[3], mul, add, _s()),
            "vd_resize_bilinear")
    return out


def channel_affine(x, mul, add, out):
    """out[b, c] = x[b, c] * mul[c] + add[c]  (contiguous NCHW)."""
    B, C, H, W = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape and mul.numel() == C and add.numel() == C
    L.check(_lib().vd_channel_affine(_p(x), _p(mul), _p(add), _p(out), B, C, H * W, _s()), "vd_channel_affine")
    return out


def lpips_layer(f0, f1, w, out, accumulate=False):
    """out[n] (+)= mean_p sum_c w[c] (unit(f0) - unit(f1))^2 for one tap of the LPIPS metric (contiguous [N, C, H, W] feature maps)."""
    N, C, H, W = f0.shape
    assert f0.is_contiguous() and f1.is_contiguous() and f1.shape == f0.shape and w.numel() == C and out.numel() == N
    L.check(_lib().vd_lpips_layer(_p(f0), _p(f1), _p(w), _p(out), N, C, H * W, int(accumulate), _s()), "vd_lpips_layer")
    return out


def attn_core_eligible(heads, head_dim, N) -> bool:
    """Shapes the fused attention core takes (vd_attn_core_fwd / _bwd): 256 tokens, head_dim 32 / 64 / 128 or a multiple of 256."""
    return N == 256 and heads >= 1 and (head_dim in (32, 64, 128) or (head_dim > 0 and head_dim % 256 == 0))


def gemm_bx3_act_eligible(M, K, NP) -> bool:
    """Products of two activation matrices (attention) the split-precision kernel takes (vd_gemm_desc.math = 1)."""
    return NP % 128 == 0 and K % 16 == 0 and K >= 32 and M >= 64 and M % 4 == 0


def gemm_bx3_eligible(M, K, NP, nb=None) -> bool:
    """1x1 convolutions / plain products the split-precision GEMM kernel takes (vd_gemm_desc.a_packed, shared A); nb = batch items."""
    if K % 16 != 0 or M < 64:
        return False
    return NP % 128 == 0 or (128 % NP == 0 and nb is not None and (nb * NP) % 128 == 0)


def conv1x1(x, w2d, bias, out, residual=None, accumulate=False, tile=0, a_packed=None):
    """1x1 convolution on NCHW = batched GEMM W[M,C] @ x[b][C, HW]."""
    Bn, Cc, H, W, xbs = _img(x)
    M = w2d.shape[0]
    assert w2d.shape[1] == Cc and w2d.is_contiguous()
    Bo, Mo, Ho, Wo, obs = _img(out)
    assert (Bo, Mo, Ho, Wo) == (Bn, M, H, W)
    rbs = _img(residual)[4] if residual is not None else 0
    HW = H * W
    return gemm(w2d, x, out, M=M, N=Bn * HW, K=Cc, b_mode=B_PLAIN, NP=HW, lda=Cc, ldb=HW, b_bstride=xbs, ldd=HW,
                d_bstride=obs, bias=bias, residual=residual, res_bstride=rbs, accumulate=accumulate, tile=tile, a_packed=a_packed)


def linear(x, w, bias, out, accumulate=False):
    """out[b, o] = sum_i x[b, i] w[o, i] + bias[o]   (x: [B, I], w: [O, I])."""
    Bn, I = x.shape
    O = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and out.shape == (Bn, O)
    return gemm(x, w, out, M=Bn, N=O, K=I, a_mode=A_ROW, b_mode=B_KCONTIG, lda=I, ldb=w.stride(0), ldd=O,
                bias=bias, bias_on_n=True, accumulate=accumulate)


def linear_dgrad(dy, w, dx, accumulate=False):
    """dx[b, i] = sum_o dy[b, o] w[o, i]."""
    Bn, O = dy.shape
    I = w.shape[1]
    assert dy.is_contiguous() and dx.is_contiguous()
    S = _long_k_slices(Bn, I, O)
    if S > 1:
        # a long reduction over few output tiles (the time-embedding projections of all ResNet blocks at once: K = 4992, 16 tiles):
        # K slices run as the GEMM's batch dimension into [S][Bn][I] partials, summed in slice order by colsum (deterministic)
        Kc = O // S
        ws = _gemm_ws(S * Bn * I, dx.device)
        gemm(dy, w, ws, M=Bn, N=S * I, K=Kc, NP=I, a_mode=A_ROW, b_mode=B_PLAIN, lda=O, a_bstride=Kc, ldb=w.stride(0),
             b_bstride=Kc * w.stride(0), ldd=I, d_bstride=Bn * I)
        return colsum(ws, dx, S, Bn * I, accumulate=accumulate)
    return gemm(dy, w, dx, M=Bn, N=I, K=O, a_mode=A_ROW, b_mode=B_PLAIN, lda=O, ldb=w.stride(0), ldd=I,
                accumulate=accumulate)


def _long_k_slices(M, N, K) -> int:
    """Number of K slices for a plain product whose tile grid cannot fill the chip by itself (0/1 = do not slice)."""
    if K < 2048 or N % 128 != 0 or ((M + 63) // 64) * (N // 64) >= 128:
        return 1
    for S in range(32, 1, -1):
        if K % S == 0 and (K // S) % 4 == 0 and K // S >= 256:
            return S
    return 1


def linear_wgrad(dy, x, dw, accumulate=False):
    """dw[o, i] (+)= sum_b dy[b, o] x[b, i]."""
    Bn, O = dy.shape
    I = x.shape[1]
    assert dy.is_contiguous() and x.is_contiguous()
    return gemm(dy, x, dw, M=O, N=I, K=Bn, a_mode=A_COL, b_mode=B_PLAIN, lda=O, ldb=I, ldd=dw.stride(0),
                accumulate=accumulate)


def wgrad_desc(dy, x, dw2d, mode, ws=None, accumulate=False, splits=0, tile=0, pad=0, math_mode=0) -> WgradDesc:
    dy, dy_ps = _unwrap(dy)                     # PreSplit operands: vd_wgrad_desc.presplit (bit 0: x, bit 1: dy)
    x, x_ps = _unwrap(x)
    Bn, M, OH, OW, dbs = _img(dy)
    Bx, Cc, H, W, xbs = _img(x)
    assert Bx == Bn
    T = 1 if mode == B_PLAIN else 9
    assert dw2d.shape == (M, Cc * T) and dw2d.is_contiguous()
    d = WgradDesc()
    d.dY, d.X, d.dW, d.ws = _p(dy), _p(x), _p(dw2d), _p(ws)
    d.M, d.C, d.T, d.nb, d.NP = M, Cc, T, Bn, OH * OW
    d.H, d.W, d.OH, d.OW = H, W, OH, OW
    d.mode, d.splits, d.accumulate, d.tile = mode, splits, int(accumulate), tile
    d.dy_bstride, d.x_bstride = dbs, xbs
    d.pad = pad
    d.math = math_mode
    d.presplit = int(x_ps) | (int(dy_ps) << 1)
    return d


def wgrad_group_class(d: WgradDesc) -> int:
    """Kernel class of a split-precision weight gradient for the grouped launch (0: not groupable -> conv_wgrad)."""
    return int(L.load().vd_conv_wgrad_group_class(C.byref(d)))


_WG_CACHE, _WG_WS = {}, {}

# Device job tables (grouped weight gradients, segmented column sums) are built on the host from operand ADDRESSES and uploaded once per
# distinct set.  Inside a HIP-graph capture the activations live at new addresses (the graph's private pool), so new tables appear -- and
# a pageable host -> device copy cannot be captured.  While a capture is open the device tensor is only ALLOCATED (from the graph's pool:
# its address is what the captured launches read) and the copy (+ any one-time fix-up kernel) runs right after the capture ends; the
# graph never writes these tables, so one upload serves every replay.
_CAPTURE_DEFER = None
_CAPTURE_TABLES = None
_CAPTURE_ARENA = None      # [uint8 device buffer, bytes used]


def capture_begin(device, arena_bytes: int = 1 << 20):
    """Open a graph capture for table uploads.  The tables must NOT come from the graph's private memory pool: a block the capture hands to
    a table may have belonged to an activation EARLIER in capture order, and every replay would then overwrite the uploaded table with that
    activation before the launches that read it run.  They are carved out of an arena allocated here, before the capture starts."""
    global _CAPTURE_DEFER, _CAPTURE_TABLES, _CAPTURE_ARENA
    _CAPTURE_DEFER, _CAPTURE_TABLES = [], []
    _CAPTURE_ARENA = [torch.empty(arena_bytes, dtype=torch.uint8, device=device), 0]


def capture_end():
    """Runs the deferred uploads; returns the device tables (and their arena) the captured launches read -- the graph's owner must keep them
    alive (the look-up caches that also hold them are bounded and may drop them)."""
    global _CAPTURE_DEFER, _CAPTURE_TABLES, _CAPTURE_ARENA
    todo, _CAPTURE_DEFER = _CAPTURE_DEFER or [], None
    tables, _CAPTURE_TABLES = _CAPTURE_TABLES or [], None
    arena, _CAPTURE_ARENA = _CAPTURE_ARENA, None
    for fin in todo:
        fin()
    return tables + ([arena[0]] if arena else [])


def upload_table(host: torch.Tensor, device, after=None) -> torch.Tensor:
    """host (CPU tensor) -> device, now or -- during a graph capture -- right after it; `after(dev_tensor)` runs once the data is there."""
    if _CAPTURE_DEFER is None:
        t = host.to(device)
        if after is not None:
            after(t)
        return t
    nbytes = host.numel() * host.element_size()
    arena, used = _CAPTURE_ARENA
    start = (used + 255) // 256 * 256
    if start + nbytes > arena.numel():
        raise L.VillanHipError(f"graph capture: job-table arena exhausted ({start + nbytes} > {arena.numel()} bytes)")
    _CAPTURE_ARENA[1] = start + nbytes
    t = arena[start:start + nbytes].view(host.dtype).view(host.shape)
    _CAPTURE_TABLES.append(t)

    def fin():
        t.copy_(host)
        if after is not None:
            after(t)
    _CAPTURE_DEFER.append(fin)
    return t


def conv_wgrad_group(descs: Sequence[WgradDesc], device):
    """All `descs` (one kernel class, see wgrad_group_class) in ONE compute launch + ONE fixed-order slab reduction.  The device job
    table is planned by the library and uploaded once per distinct set of operand addresses (steady-state training repeats them)."""
    lib = _lib()
    n = len(descs)
    key = tuple((d.dY, d.X, d.dW, d.M, d.C, d.T, d.nb, d.NP, d.H, d.W, d.OH, d.OW, d.mode, d.accumulate, d.dy_bstride, d.x_bstride, d.presplit)
                for d in descs)
    ent = _WG_CACHE.get(key)
    ws = _WG_WS.get(device)
    if ent is None or ws is None or ent["ws_ptr"] != ws.data_ptr() or ent["ws_floats"] > ws.numel():
        arr = (WgradDesc * n)(*descs)
        jb = int(lib.vd_conv_wgrad_group_job_bytes())
        host = (C.c_uint8 * (n * jb))()
        wsf, blocks, rblocks = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        cls = lib.vd_conv_wgrad_group_plan(arr, n, host, C.byref(wsf), C.byref(blocks), C.byref(rblocks))
        if cls <= 0:
            raise L.VillanHipError(f"vd_conv_wgrad_group_plan: {lib.vd_last_error().decode()}")
        if ws is None or ws.numel() < wsf.value:
            ws = torch.empty(max(int(wsf.value), 1 << 22), device=device, dtype=torch.float32)
            _WG_WS[device] = ws
        ws_ptr = ws.data_ptr()
        table = upload_table(torch.frombuffer(bytearray(host), dtype=torch.uint8), device,
                             after=lambda t: L.check(lib.vd_conv_wgrad_group_rebase(t.data_ptr(), n, ws_ptr, _s()), "vd_conv_wgrad_group_rebase"))
        if len(_WG_CACHE) > 256:
            _WG_CACHE.clear()
        ent = _WG_CACHE[key] = {"table": table, "cls": cls, "blocks": blocks.value, "rblocks": rblocks.value, "ws_ptr": ws.data_ptr(),
                                "ws_floats": int(wsf.value)}
    if _CAPTURE_TABLES is not None:
        _CAPTURE_TABLES.append(ent["table"])              # a cache hit inside a capture: the graph reads this table too
    flops = sum(2.0 * d.M * d.C * d.T * d.nb * d.NP for d in descs)
    nbytes = sum(4.0 * (d.nb * d.M * d.NP + d.nb * d.C * d.H * d.W + d.M * d.C * d.T) for d in descs)
    var = lib.vd_conv_wgrad_group_variant(ent["cls"])
    if ent["cls"] >= 3000:                                       # both operands pre-split: LDS-DMA + transposed reads (vd_presplit.hip)
        name = f"wgrad_ps_group_kernel<{(ent['cls'] - 3000) // 4}, {(ent['cls'] - 3000) & 2}>(+group_reduce)"
    elif ent["cls"] > 2000:                                      # stride-2 3x3 classes (2000 + output width; 2033: 32-pixel segments of wide outputs)
        name = "wgrad_bx3_group_kernel<32, 4, true>(+group_reduce)" if ent["cls"] == 2033 else f"wgrad_bx3_group_kernel<{ent['cls'] - 2000}, 4, false>(+group_reduce)"
    elif ent["cls"] == 1000:                                     # symbol names as rocprofv3 prints them
        name = "wgrad1x1_wide_group_kernel(+group_reduce)" if var == 256 else "wgrad1x1_bx3_group_kernel(+group_reduce)"
    elif var == 9:
        name = f"wgrad9_group_kernel<{ent['cls'] // 4}>(+group_reduce)"
    elif var == 32:
        name = f"wgrad_k32_group_kernel<{ent['cls'] // 4}, {ent['cls'] & 2}>(+group_reduce)"
    else:
        name = f"wgrad_bx3_group_kernel<{ent['cls'] // 4}, {ent['cls'] & 2}, {'true' if ent['cls'] & 1 else 'false'}>(+group_reduce)"
    _timed(name, flops, "mfma", lambda: L.check(
        lib.vd_conv_wgrad_group_launch(ent["table"].data_ptr(), n, ent["cls"], ent["blocks"], ent["rblocks"], _s()), "vd_conv_wgrad_group_launch"),
        nbytes=nbytes)


def conv_wgrad(dy, x, dw2d, mode, ws: Optional[torch.Tensor], accumulate=False, splits=0, tile=0, pad=0, math_mode=0):
    """dw2d[M, C*T] (+)= sum_{b,p} dy[b,m,p] * gather_mode(x)[b, c, p(+)t]."""
    d = wgrad_desc(dy, x, dw2d, mode, ws, accumulate, splits, tile, pad, math_mode)
    Bn, M, OH, OW = dy.shape
    Cc, T = x.shape[1], d.T
    lib = _lib()
    need = lib.vd_conv_wgrad_ws_floats(C.byref(d))
    if need > 0:
        assert ws is not None and ws.numel() >= need, f"wgrad workspace too small: need {need}"
    if _PROF is None:
        L.check(lib.vd_conv_wgrad(C.byref(d), _s()), "vd_conv_wgrad")
        return dw2d
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.vd_conv_wgrad(C.byref(d), _s()), "vd_conv_wgrad")
    e1.record()
    tl, sp = C.c_int32(0), C.c_int32(0)
    lib.vd_conv_wgrad_plan(C.byref(d), C.byref(tl), C.byref(sp))
    if tl.value == 6:
        name = f"wgrad_small_kernel<{'true' if M <= 4 else 'false'}>(+colsum)"
    elif tl.value == 5:
        name = "wgrad1x1_bx3_kernel(+slab_reduce)"
    elif tl.value == 4 and math_mode == 1:
        name = f"wgrad_bx3_kernel<{min(OW, 32)}, {0 if mode == B_CONV3 else 2}{', wide' if OW > 32 else ''}>(+slab_reduce)"
    elif tl.value == 4:
        name = f"wgrad_patch_kernel<{OW}, {0 if mode == B_CONV3 else 2}>(+slab_reduce)"
    else:
        name = f"wgrad_kernel<{_TILE_NAMES[tl.value]},{_B_NAMES[mode]}>(+slab_reduce)"
    _PROF.append({"name": name, "flops": 2.0 * M * Cc * T * Bn * OH * OW, "bytes": 4.0 * (dy.numel() + x.numel() + M * Cc * T),
                  "e0": e0, "e1": e1, "kind": "mfma"})
    return dw2d


def _wgrad_s2_split(OW) -> bool:
    """Stride-2 weight gradients on the split-precision kernel (wgrad_bx3 MODE 4).  VILLAN_WGRAD_S2=none: the exact-f32 kernel; wide: only 32x32 outputs and
    wider.  (The first version of the kernel measured neutral -- profiles/r04_wgrad_s2_ab.txt -- because a conditional store pattern kept its loader state in
    scratch memory; fixed, it is 0.28 ms against 0.99 on config #4's two large layers and -0.03 ms per config-#2 step: profiles/r04_wgrad_s2_wide_ab.txt.)"""
    sel = os.environ.get("VILLAN_WGRAD_S2", "all")
    return sel == "all" or (sel != "none" and OW >= 32)


def wgrad_bx3_eligible(M, Cc, OH, OW, mode) -> bool:
    """Problems the split-precision weight-gradient kernel takes (vd_wgrad_desc.math = 1)."""
    if mode == B_PLAIN:
        return (OH * OW) % 8 == 0 and M >= 64 and Cc >= 64
    if (mode == B_CONV3 and OW == 4 and OH == 4) or (mode in (B_CONV3, B_CONV3_UP) and OW >= 64 and OW % 32 == 0):
        return M >= 64 and Cc >= 64      # 4x4: two images per K-step; wide images: 32-pixel row segments
    if mode == B_CONV3_S2:                   # round 4: the Downsample2D convolution (stride 2) at 8x8 .. 32x32 outputs and 32-pixel segments of wider ones
        return OH == OW and (OW in (8, 16, 32) or (OW >= 64 and OW % 32 == 0)) and M >= 64 and Cc >= 64 and _wgrad_s2_split(OW)
    return mode in (B_CONV3, B_CONV3_UP) and OH == OW and OW in (8, 16, 32) and M >= 64 and Cc >= 64


def wgrad_ws_floats(M, Cc, T, nb, NP, OH=None, OW=None, mode=None, math_mode=0) -> int:
    """Workspace floats vd_conv_wgrad needs (square outputs assumed when OH/OW are omitted)."""
    d = WgradDesc()
    if OH is None:
        OH = OW = int(round(math.sqrt(NP)))
    d.M, d.C, d.T, d.nb, d.NP, d.OH, d.OW = M, Cc, T, nb, NP, OH, OW
    d.mode = (B_PLAIN if T == 1 else B_CONV3) if mode is None else mode
    d.math = math_mode
    scale = {B_CONV3_UP: (1, 2), B_CONV3_S2: (2, 1)}.get(d.mode, (1, 1))          # source dims of the gather
    d.H, d.W = OH * scale[0] // scale[1], OW * scale[0] // scale[1]
    return int(L.load().vd_conv_wgrad_ws_floats(C.byref(d)))


def weight_transpose(w2d, wt, M, Cc, T):
    L.check(_lib().vd_weight_transpose(_p(w2d), _p(wt), M, Cc, T, _s()), "vd_weight_transpose")
    return wt


def sumpool2x2(dU, dX, accumulate=False):
    Bn, Cc, H, W, xbs = _img(dX)
    ubs = _img(dU)[4]
    assert dU.shape == (Bn, Cc, 2 * H, 2 * W)
    L.check(_lib().vd_sumpool2x2(_p(dU), _p(dX), Bn, Cc, H, W, ubs, xbs, int(accumulate), _s()), "vd_sumpool2x2")
    return dX


def conv3x3_s2_dgrad(dy, w2d, dx, pad=0, a_packed=None):
    """Input gradient of the stride-2 (pad (0,1,0,1)) conv: plain GEMM G[b] = w2d^T @ dy[b] (no structural zeros, unlike
    the dilated-gather GEMM), then the col2im gather.  a_packed: the [C*9, M] transpose packed as a one-tap split-precision operand."""
    Bn, M, OH, OW, dbs = _img(dy)
    Bx, Cc, H, W, xbs = _img(dx)
    assert Bx == Bn and w2d.shape == (M, Cc * 9) and w2d.is_contiguous()
    OHW = OH * OW
    G = torch.empty((Bn, Cc * 9, OHW), device=dy.device, dtype=torch.float32)
    gemm(w2d, dy, G, M=Cc * 9, N=Bn * OHW, K=M, a_mode=A_COL, b_mode=B_PLAIN, NP=OHW, lda=Cc * 9, ldb=OHW, b_bstride=dbs,
         ldd=OHW, d_bstride=Cc * 9 * OHW, a_packed=a_packed)
    L.check(_lib().vd_col2im_s2(_p(G), _p(dx), Bn, Cc, H, W, OH, OW, pad, Cc * 9 * OHW, xbs, _s()), "vd_col2im_s2")
    return dx


def rowsum(x, ws, ws_ld=None):
    """ws[b, m] = sum_p x[b, m, :, :]   (ws may be a column slice of a wider [B, ld] matrix)."""
    Bn, M, H, W, xbs = _img(x)
    ld = M if ws_ld is None else ws_ld
    L.check(_lib().vd_rowsum(_p(x), _p(ws), Bn, M, H * W, xbs, ld, _s()), "vd_rowsum")
    return ws


def colsum(ws, out, Bn, Cc, ld=None, accumulate=False):
    L.check(_lib().vd_colsum(_p(ws), _p(out), Bn, Cc, Cc if ld is None else ld, int(accumulate), _s()), "vd_colsum")
    return out


def colsum_segmented(table, n_jobs, Bn):
    """table: device int64 [n_jobs, 4] = (ws address, out address, columns <= 64, ld); out += column sums (one launch)."""
    assert table.dtype == torch.int64 and table.is_contiguous() and table.numel() >= 4 * n_jobs
    L.check(_lib().vd_colsum_segmented(_p(table), n_jobs, Bn, _s()), "vd_colsum_segmented")


# --------------------------------------------------------------------------------------------- GroupNorm
def _gn_ws(Bn, Cc, HW, G, device):
    """Scratch for the multi-workgroup GroupNorm of large groups (None when the single-workgroup kernels apply)."""
    need = _lib().vd_groupnorm_ws_floats(Bn, Cc, HW, G)
    return _gemm_ws(need, device) if need > 0 else None


def groupnorm_fwd(x, gamma, beta, y, mean, rstd, G, eps, silu):
    Bn, Cc, H, W, xbs = _img(x)
    ybs = _img(y)[4]
    assert y.shape == x.shape and mean.numel() >= Bn * G and rstd.numel() >= Bn * G
    _timed("groupnorm_fwd (gn_fwd_reg_kernel<*> / gn_chunk_*)", 8.0 * x.numel(), "hbm", lambda: L.check(
        _lib().vd_groupnorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), Bn, Cc, H * W, G, eps,
                                int(silu), xbs, ybs, _p(_gn_ws(Bn, Cc, H * W, G, x.device)), _s()), "vd_groupnorm_fwd"))
    return y


def groupnorm_stats(x, gamma, beta, ss, mean, rstd, G, eps):
    """ss[b, c] = (gamma_c * rstd, beta_c - mean * gamma_c * rstd): the operand of conv3x3(gn_ss=...)."""
    Bn, Cc, H, W, xbs = _img(x)
    assert ss.shape == (Bn, Cc, 2) and ss.is_contiguous()
    L.check(_lib().vd_groupnorm_stats(_p(x), _p(gamma), _p(beta), _p(ss), _p(mean), _p(rstd), Bn, Cc, H * W, G, eps, xbs, _s()),
            "vd_groupnorm_stats")
    return ss


def groupnorm_stats_from_partials(part, tiles, gamma, beta, ss, mean, rstd, HW, G, eps):
    """groupnorm_stats() from the per-tile channel sums a convolution wrote (conv3x3(gn_part=...)): no pass over the tensor."""
    Bn, Cc = ss.shape[0], ss.shape[1]
    assert ss.is_contiguous() and part.is_contiguous() and part.numel() >= Bn * tiles * Cc * 2
    L.check(_lib().vd_groupnorm_stats_from_partials(_p(part), tiles, _p(gamma), _p(beta), _p(ss), _p(mean), _p(rstd), Bn, Cc, HW, G, eps, _s()),
            "vd_groupnorm_stats_from_partials")
    return ss


def gn_fusable(x, cout) -> bool:
    """conv3x3(gn_ss=...) applies: patch-staged kernel with one image per tile and a register-resident statistics pass."""
    Bn, Cc, H, W = x.shape
    return W in (16, 32) and H == W and Cc % 8 == 0 and Cc <= 1024 and cout >= 64


def groupnorm_bwd(dy, x, mean, rstd, gamma, beta, dx, dgamma_ws, dbeta_ws, G, silu, extra=None, extra2=None, rowsum=None, rowsum_ld=None):
    """dx = GN(+SiLU) backward (+ extra + extra2); rowsum (optional, [B, >= C] rows of stride rowsum_ld): sum_p dx[b, c, p], written by the
    same pass (vd_groupnorm_bwd_fused)."""
    Bn, Cc, H, W, xbs = _img(x)
    dbs, dxbs = _img(dy)[4], _img(dx)[4]
    ebs = _img(extra)[4] if extra is not None else 0
    e2bs = _img(extra2)[4] if extra2 is not None else 0
    assert dgamma_ws.numel() >= Bn * Cc and dbeta_ws.numel() >= Bn * Cc
    assert extra2 is None or extra2.shape == x.shape
    rld = 0
    if rowsum is not None:
        rld = Cc if rowsum_ld is None else rowsum_ld
        assert rld >= Cc
    nbytes = (12.0 + (4.0 if extra is not None else 0.0) + (4.0 if extra2 is not None else 0.0)) * x.numel()   # dy, x (+ extras) read once, dx written once
    _timed("groupnorm_bwd (gn_bwd_reg_kernel<*> / gn_chunk_*)", nbytes, "hbm", lambda: L.check(
        _lib().vd_groupnorm_bwd_fused(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(extra), _p(extra2), _p(dx),
                                      _p(dgamma_ws), _p(dbeta_ws), _p(rowsum), Bn, Cc, H * W, G, int(silu), dbs, xbs, ebs, e2bs, dxbs, rld,
                                      _p(_gn_ws(Bn, Cc, H * W, G, x.device)), _s()), "vd_groupnorm_bwd"))
    return dx


# --------------------------------------------------------------------------------------------- attention
def softmax_col_fwd(S, nb, N):
    L.check(_lib().vd_softmax_col_fwd(_p(S), nb, N, _s()), "vd_softmax_col_fwd")
    return S


def softmax_col_bwd(P, dP, nb, N, scale):
    L.check(_lib().vd_softmax_col_bwd(_p(P), _p(dP), nb, N, scale, _s()), "vd_softmax_col_bwd")
    return dP


def attn_core_fwd(qkv, out, P, heads, head_dim, N, scale):
    """Fused attention core (qkv [B, 3C, H, W] contiguous); P = None in the no-grad path, else the [B, heads, N, N] probabilities."""
    Bn = qkv.shape[0]
    assert qkv.is_contiguous() and out.is_contiguous() and (P is None or P.is_contiguous())
    dt = 8 if head_dim % 256 == 0 else head_dim // 32
    _timed(f"attn_core_kernel<{dt}, false>", 4.0 * Bn * heads * N * N * head_dim, "mfma", lambda: L.check(
        _lib().vd_attn_core_fwd(_p(qkv), _p(out), _p(P), Bn, heads, head_dim, N, scale, _s()), "vd_attn_core_fwd"),
        nbytes=4.0 * (qkv.numel() + out.numel() + (P.numel() if P is not None else 0)))
    return out


def attn_core_bwd(qkv, P, out, dout, dS, dqkv, heads, head_dim, N, scale):
    """dS (into `dS`, [B, heads, N, N]) and dq (into dqkv[:, :C]) of the fused attention core; `out` = the saved forward output."""
    Bn = qkv.shape[0]
    assert qkv.is_contiguous() and P.is_contiguous() and dout.is_contiguous() and dS.is_contiguous() and dqkv.is_contiguous() and out.is_contiguous()
    dt = 8 if head_dim % 256 == 0 else head_dim // 32
    _timed(f"attn_core_kernel<{dt}, true>", 4.0 * Bn * heads * N * N * head_dim, "mfma", lambda: L.check(
        _lib().vd_attn_core_bwd(_p(qkv), _p(P), _p(out), _p(dout), _p(dS), _p(dqkv), Bn, heads, head_dim, N, scale, _s()), "vd_attn_core_bwd"),
        nbytes=4.0 * (qkv.numel() * 2 // 3 + 2 * dout.numel() + P.numel() + dS.numel() + qkv.numel() // 3))
    return dqkv


def attn_flash_eligible(heads, head_dim, N) -> bool:
    """Shapes the flash attention takes (vd_attn_flash_fwd / _bwd): heads of 32 channels and a multiple of 256 tokens (at 256 tokens it
    replaces attn_core + the dk / dv products: no probability matrix is saved)."""
    return head_dim == 32 and heads >= 1 and N >= 256 and N % 256 == 0


def attn_flash_fwd(qkv, out, lse, heads, head_dim, N, scale):
    """Attention of qkv [B, 3C, H, W] without the score matrix in HBM; lse [B, heads, N] (None in the no-grad path) for the backward pass."""
    Bn = qkv.shape[0]
    assert qkv.is_contiguous() and out.is_contiguous() and (lse is None or lse.is_contiguous())
    _timed("attn_flash_kernel<0>", 4.0 * Bn * heads * N * N * head_dim, "mfma", lambda: L.check(
        _lib().vd_attn_flash_fwd(_p(qkv), _p(out), _p(lse), Bn, heads, head_dim, N, scale, qkv.stride(0), out.stride(0), _s()), "vd_attn_flash_fwd"),
        nbytes=4.0 * (qkv.numel() + out.numel()))
    return out


def attn_flash_bwd(qkv, out, dout, lse, dqkv, heads, head_dim, N, scale):
    """dq, dk, dv (all of dqkv [B, 3C, H, W]) from the saved forward output and lse; P is recomputed block by block."""
    Bn = qkv.shape[0]
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and lse.is_contiguous() and dqkv.is_contiguous()
    delta = torch.empty_like(lse)
    _timed("attn_flash_kernel<1>+<2>", 10.0 * Bn * heads * N * N * head_dim, "mfma", lambda: L.check(
        _lib().vd_attn_flash_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), Bn, heads, head_dim, N, scale, qkv.stride(0),
                                 out.stride(0), dout.stride(0), dqkv.stride(0), _s()), "vd_attn_flash_bwd"),
        nbytes=4.0 * (2 * qkv.numel() + 2 * dout.numel()))
    return dqkv


def attn_small_fwd(qkv, out, P, Cc, N, scale):
    Bn = qkv.shape[0]
    L.check(_lib().vd_attn_small_fwd(_p(qkv), _p(out), _p(P), Bn, Cc, N, scale, qkv.stride(0), out.stride(0), _s()),
            "vd_attn_small_fwd")
    return out


def attn_small_bwd(qkv, P, dout, dqkv, Cc, N, scale):
    Bn = qkv.shape[0]
    L.check(_lib().vd_attn_small_bwd(_p(qkv), _p(P), _p(dout), _p(dqkv), Bn, Cc, N, scale, qkv.stride(0),
                                     dout.stride(0), dqkv.stride(0), _s()), "vd_attn_small_bwd")
    return dqkv


# --------------------------------------------------------------------------------------------- elementwise
def timestep_embedding(t_f32, freqs, emb, flip):
    Bn, half = t_f32.numel(), freqs.numel()
    assert emb.shape == (Bn, 2 * half) and emb.is_contiguous()
    L.check(_lib().vd_timestep_embedding(_p(t_f32), _p(freqs), _p(emb), Bn, half, int(flip), _s()),
            "vd_timestep_embedding")
    return emb


def silu_fwd(x, y):
    L.check(_lib().vd_silu_fwd(_p(x), _p(y), x.numel(), _s()), "vd_silu_fwd")
    return y


def silu_bwd(dy, x, dx, accumulate=False):
    L.check(_lib().vd_silu_bwd(_p(dy), _p(x), _p(dx), x.numel(), int(accumulate), _s()), "vd_silu_bwd")
    return dx


def add_strided(dst, src, accumulate=True):
    """dst (+)= src for image views with (possibly different) batch strides."""
    Bn, Cc, H, W, dbs = _img(dst)
    sbs = _img(src)[4]
    assert src.shape == dst.shape
    L.check(_lib().vd_add_strided(_p(dst), _p(src), Bn, Cc * H * W, dbs, sbs, int(accumulate), _s()), "vd_add_strided")
    return dst


def add_flat(dst, src, accumulate=True):
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    L.check(_lib().vd_add_strided(_p(dst), _p(src), 1, dst.numel(), dst.numel(), src.numel(), int(accumulate), _s()),
            "vd_add_strided")
    return dst


def scale_(x, alpha: float):
    assert x.is_contiguous()
    L.check(_lib().vd_scale(_p(x), x.numel(), alpha, _s()), "vd_scale")
    return x


def lincomb(out, srcs: Sequence[torch.Tensor], coefs: Sequence[float]):
    n = len(srcs)
    assert 1 <= n <= 6 and len(coefs) == n
    for t in srcs:
        assert t.is_contiguous() and t.numel() == out.numel()
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    L.check(_lib().vd_lincomb(_p(out), ptrs, cf, n, out.numel(), _s()), "vd_lincomb")
    return out


# --------------------------------------------------------------------------------------------- loss / optimiser
def qsample_backdoor(x0, R, eps, t, tab_a, tab_s, tab_step, tab_coef, x_t, y):
    Bn = x0.shape[0]
    chw = x0.numel() // Bn
    for z in (x0, R, eps, x_t, y):
        assert z.is_contiguous() and z.dtype == torch.float32
    assert t.dtype == torch.int64 and t.is_contiguous()
    L.check(_lib().vd_qsample_backdoor(_p(x0), _p(R), _p(eps), _p(t), _p(tab_a), _p(tab_s), _p(tab_step), _p(tab_coef),
                                       _p(x_t), _p(y), Bn, chw, _s()), "vd_qsample_backdoor")
    return x_t, y


LOSS_KINDS = {"l2": 0, "l1": 1, "huber": 2}


def mse_fwd_bwd(pred, y, dpred, loss, partial, pscale=None, gscale=1.0, kind="l2"):
    """loss = mean(norm(pred * pscale[b] - y)) and its gradient in one pass; kind = the reference's loss_type (loss.py:849-858)."""
    Bn = pred.shape[0]
    chw = pred.numel() // Bn
    assert pred.is_contiguous() and y.is_contiguous() and partial.numel() >= 1024
    L.check(_lib().vd_loss_fwd_bwd(_p(pred), _p(y), _p(pscale), _p(dpred), _p(loss), _p(partial), Bn, chw, gscale,
                                   LOSS_KINDS[kind], _s()), "vd_loss_fwd_bwd")
    return loss


def l2norm_sq(g, partial, out_sq):
    assert g.is_contiguous() and partial.numel() >= 1024
    _timed("l2norm_sq (sumsq_kernel)", 4.0 * g.numel(), "hbm",
           lambda: L.check(_lib().vd_l2norm_sq(_p(g), g.numel(), _p(partial), _p(out_sq), _s()), "vd_l2norm_sq"))
    return out_sq


def adam_step(p, g, m, v, norm_sq, max_norm, inv_scale, lr, beta1, beta2, eps, step, skipped=None):
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1
    _timed("adam_step (adam_kernel)", 28.0 * p.numel(), "hbm", lambda: L.check(      # p, g, m, v read; p, m, v written
        _lib().vd_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(norm_sq), max_norm, inv_scale, lr, beta1, beta2,
                            eps, step, _p(skipped), _s()), "vd_adam_step"))


# --------------------------------------------------------------------------------------------- samplers / data
def sched_step(x, eps, out, *, c_eps, c_div, clip, c_x0, c_x, c_e, c_z, z=None, x0_out=None, seed=0, offset=0):
    assert x.is_contiguous() and eps.is_contiguous() and out.is_contiguous()
    L.check(_lib().vd_sched_step(_p(x), _p(eps), _p(z), _p(out), _p(x0_out), x.numel(), c_eps, c_div, clip, c_x0, c_x, c_e,
                                 c_z, seed, offset, _s()), "vd_sched_step")
    return out


def batch_l2norm(x, out):
    """out[b] = ||x[b]||_2 over all non-batch dims."""
    Bn = x.shape[0]
    assert x.is_contiguous() and out.numel() >= Bn
    L.check(_lib().vd_batch_l2norm(_p(x), _p(out), Bn, x.numel() // Bn, _s()), "vd_batch_l2norm")
    return out


def ssim(a, b, win, out, c1, c2):
    """out[n] = mean SSIM map of (a[n], b[n]); win: [K, K] window (vd_ssim)."""
    Bn, Cc, H, W = a.shape
    assert a.is_contiguous() and b.is_contiguous() and win.is_contiguous() and a.shape == b.shape and win.shape[0] == win.shape[1]
    L.check(_lib().vd_ssim(_p(a), _p(b), _p(win), _p(out), Bn, Cc, H, W, win.shape[0], c1, c2, _s()), "vd_ssim")
    return out


def postprocess(x, out, mul, add, lo, hi, to_nhwc):
    Bn, Cc, H, W = x.shape
    assert x.is_contiguous() and out.is_contiguous()
    L.check(_lib().vd_postprocess(_p(x), _p(out), Bn, Cc, H * W, mul, add, lo, hi, int(to_nhwc), _s()), "vd_postprocess")
    return out


def fir_resample2(x, out, up: bool, scale: float = 1.0, accumulate: bool = False):
    """diffusers upsample_2d / downsample_2d ((1,3,3,1) FIR, factor 2) on contiguous [B, C, H, W]; out = scale * f(x) (+ out)."""
    Bn, Cc, H, W = x.shape
    assert x.is_contiguous() and out.is_contiguous()
    assert out.shape == ((Bn, Cc, 2 * H, 2 * W) if up else (Bn, Cc, H // 2, W // 2)), (x.shape, out.shape, up)
    L.check(_lib().vd_fir_resample2(_p(x), _p(out), Bn * Cc, H, W, int(up), scale, int(accumulate), _s()), "vd_fir_resample2")
    return out


def fourier_embedding(t_f32, W, emb):
    Bn, half = t_f32.numel(), W.numel()
    assert emb.shape == (Bn, 2 * half) and emb.is_contiguous()
    L.check(_lib().vd_fourier_embedding(_p(t_f32), _p(W), _p(emb), Bn, half, _s()), "vd_fourier_embedding")
    return emb


def rowscale(x, s, out, divide=False):
    Bn = x.shape[0]
    assert x.is_contiguous() and out.is_contiguous() and s.numel() == Bn
    L.check(_lib().vd_rowscale(_p(x), _p(s), _p(out), Bn, x.numel() // Bn, int(divide), _s()), "vd_rowscale")
    return out


def vq_nearest(z, codebook, zq, idx=None):
    """zq[b, :, p] = codebook[argmin_e |z[b, :, p] - e|^2]  (VectorQuantizer forward)."""
    Bn, D, H, W, zbs = _img(z)
    qbs = _img(zq)[4]
    assert zq.shape == z.shape and codebook.is_contiguous() and codebook.shape[1] == D
    assert idx is None or (idx.dtype == torch.int64 and idx.numel() >= Bn * H * W)
    L.check(_lib().vd_vq_nearest(_p(z), _p(codebook), _p(zq), _p(idx), Bn, D, H * W, codebook.shape[0], zbs, qbs, _s()),
            "vd_vq_nearest")
    return zq


def randn(out, seed, offset):
    L.check(_lib().vd_randn(_p(out), out.numel(), seed, offset, _s()), "vd_randn")
    return out


def poison_batch(img_u8, flags_u8, trigger, target, pixel_values, tgt, image_out, vmin, vmax, R_trigger_only=False, idx=None):
    """img_u8: the [N, H, W, C] uint8 dataset (or exactly the batch when idx is None); idx: int64 [B] sample ids."""
    _, H, W, Cc = img_u8.shape
    Bn = flags_u8.numel()
    assert img_u8.dtype == torch.uint8 and flags_u8.dtype == torch.uint8 and img_u8.is_contiguous()
    assert idx is None or (idx.dtype == torch.int64 and idx.numel() == Bn and idx.is_contiguous())
    assert idx is not None or img_u8.shape[0] == Bn
    L.check(_lib().vd_poison_batch(_p(img_u8), _p(idx), _p(flags_u8), _p(trigger), _p(target), _p(pixel_values), _p(tgt),
                                   _p(image_out), Bn, Cc, H, W, vmin, vmax, int(R_trigger_only), _s()), "vd_poison_batch")
    return pixel_values, tgt
