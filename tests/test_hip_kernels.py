"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle / plain torch fp32 on the same
seeded inputs.  Tolerances: bit-exact for integer/byte work and for the short op sequences that are restated with
individually rounded ops (q-sample, scheduler step, normalisation); 2e-5 relative-to-scale for fp32 contractions
(summation order differs from the CPU's), 1e-4 for the split-precision bf16 kernels -- all far inside north_star's 1e-3."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import (A_COL, A_ROW, B_CONV3, B_CONV3_DIL, B_CONV3_S2, B_CONV3_T, B_CONV3_UP,  # noqa: E402
                                     B_KCONTIG, B_PLAIN)

DEV = "cuda"


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(a, b, tol, what=""):
    e = rel_err(a, b)
    print(f"[parity] {what}: rel_err={e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{what}: rel_err {e:.3e} > {tol:.1e}"


def g(seed):
    return torch.Generator().manual_seed(seed)


def ref_conv(x, w, b, mode):
    if mode == B_CONV3:
        return F.conv2d(x, w, b, padding=1)
    if mode == B_CONV3_S2:
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    if mode == B_CONV3_UP:
        return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    raise ValueError(mode)


CONV_CASES = [
    # B, Cin, Cout, H, mode, tile
    (4, 128, 128, 32, B_CONV3, 1), (2, 256, 128, 16, B_CONV3, 2), (8, 64, 256, 4, B_CONV3, 3), (3, 3, 128, 32, B_CONV3, 0),
    (2, 128, 3, 32, B_CONV3, 0), (2, 96, 80, 8, B_CONV3, 0), (2, 128, 128, 16, B_CONV3_S2, 0), (2, 64, 64, 8, B_CONV3_UP, 0),
    (5, 40, 72, 16, B_CONV3, 1),
    # tile=0 -> the patch-staged kernel where eligible (C % 8 == 0, W in {16, 32}, M >= 64)
    (4, 128, 128, 32, B_CONV3, 0), (2, 256, 256, 16, B_CONV3, 0), (2, 384, 128, 32, B_CONV3, 0), (3, 512, 256, 16, B_CONV3, 0),
    (2, 128, 192, 16, B_CONV3, 0), (2, 256, 256, 16, B_CONV3_UP, 0), (3, 64, 72, 32, B_CONV3, 0),
    # 8x8 / 4x4: multi-image tiles + deterministic split-K over the channel loop (ragged batch: 5 images, tile = 2 / 8 images)
    (5, 256, 256, 8, B_CONV3, 0), (16, 512, 256, 4, B_CONV3, 0), (5, 256, 128, 4, B_CONV3, 0), (3, 256, 256, 4, B_CONV3_UP, 0),
    (128, 256, 256, 4, B_CONV3, 0),
    # images wider than 32 px: row-segment tiles (64-wide: 2 rows, >= 128-wide: 128-pixel segments)
    (2, 64, 128, 64, B_CONV3, 0), (1, 72, 64, 128, B_CONV3, 0), (1, 64, 64, 256, B_CONV3, 0), (1, 64, 96, 64, B_CONV3_UP, 0),
    (1, 64, 64, 32, B_CONV3_UP, 0),
    # stride-2 (Downsample2D) through the patch-staged kernel: 32 -> 16, 16 -> 8 (2 images / tile), 8 -> 4 (8 images / tile, ragged)
    (4, 128, 128, 32, B_CONV3_S2, 0), (3, 256, 256, 16, B_CONV3_S2, 0), (5, 256, 256, 8, B_CONV3_S2, 0), (2, 72, 64, 32, B_CONV3_S2, 0),
]


@pytest.mark.parametrize("B,Cin,Cout,H,mode,tile", CONV_CASES)
def test_conv3x3_forward_epilogue(B, Cin, Cout, H, mode, tile):
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g(2))
    temb = torch.randn(B, Cout + 5, generator=g(3))
    y_ref = ref_conv(x, w, b, mode)
    res = torch.randn(y_ref.shape, generator=g(4))
    y_ref = y_ref + temb[:, 2:2 + Cout, None, None] + res
    OH = y_ref.shape[-1]
    # strided input and output views (channel slices of wider buffers)
    xbuf = torch.zeros(B, Cin + 3, H, H, device=DEV)
    xbuf[:, 3:] = x.to(DEV)
    obuf = torch.full((B, Cout + 2, OH, OH), 7.0, device=DEV)
    ops.conv3x3(xbuf[:, 3:], w.to(DEV).view(Cout, -1), b.to(DEV), obuf[:, 1:1 + Cout], mode=mode,
                rowadd=temb.to(DEV)[:, 2:], rowadd_bstride=Cout + 5, residual=res.to(DEV), tile=tile)
    check(obuf[:, 1:1 + Cout], y_ref, 2e-5, f"conv3x3 mode={mode} {Cin}->{Cout}@{H} tile={tile}")
    assert float((obuf[:, 0] - 7).abs().max()) == 0 and float((obuf[:, -1] - 7).abs().max()) == 0  # no out-of-slice writes


@pytest.mark.parametrize("B,Cin,Cout,H,mode", [(2, 128, 256, 16, B_CONV3), (4, 256, 256, 4, B_CONV3), (2, 64, 64, 16, B_CONV3_S2),
                                               (9, 256, 256, 8, B_CONV3), (20, 128, 256, 4, B_CONV3),
                                               (8, 128, 200, 8, B_CONV3), (6, 96, 128, 4, B_CONV3), (3, 128, 64, 4, B_CONV3),
                                               (1, 64, 128, 64, B_CONV3), (2, 72, 64, 128, B_CONV3), (2, 64, 64, 32, B_CONV3_UP),
                                               (4, 128, 128, 4, B_CONV3_UP), (2, 64, 64, 2, B_CONV3_UP),
                                               (2, 128, 128, 32, B_CONV3), (2, 200, 128, 32, B_CONV3), (2, 128, 128, 16, B_CONV3_UP),
                                               (2, 64, 96, 8, B_CONV3_UP), (2, 3, 128, 32, B_CONV3), (2, 128, 3, 32, B_CONV3),
                                               # direct kernel for a <= 4-channel side (conv_in / conv_out), ragged channel counts, 16x16, one image
                                               (5, 3, 72, 32, B_CONV3), (3, 100, 4, 16, B_CONV3), (1, 1, 32, 32, B_CONV3),
                                               (2, 3, 64, 80, B_CONV3), (1, 96, 3, 48, B_CONV3)])         # larger images: 32 x 32 tiles, ragged edges
def test_conv3x3_backward(B, Cin, Cout, H, mode):
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    y = ref_conv(x, w, None, mode)
    dy = torch.randn(y.shape, generator=g(2))
    y.backward(dy)
    OH = y.shape[-1]
    dyd, xd, wd = dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV)
    # weight gradient (split-K + deterministic slab reduction), accumulate on top of a known value
    dw = torch.full((Cout, Cin * 9), 0.5, device=DEV)
    need = ops.wgrad_ws_floats(Cout, Cin, 9, B, OH * OH)
    ws = torch.empty(max(need, 4), device=DEV)
    ops.conv_wgrad(dyd, xd, dw, mode, ws, accumulate=True)
    check(dw - 0.5, w.grad.view(Cout, -1), 3e-5, f"wgrad mode={mode} {Cin}->{Cout}@{H} (splits ws={need})")
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(dyd, xd, dw2, mode, ws, accumulate=False, splits=1)
    check(dw2, w.grad.view(Cout, -1), 3e-5, "wgrad splits=1")
    # input gradient through the transposed weights
    wt = torch.empty(Cin, Cout * 9, device=DEV)
    ops.weight_transpose(wd, wt, Cout, Cin, 9)
    assert torch.equal(wt.view(Cin, Cout, 9).cpu(), w.detach().view(Cout, Cin, 9).permute(1, 0, 2))
    dx = torch.empty(B, Cin, H, H, device=DEV)
    if mode == B_CONV3:
        ops.conv3x3(dyd, wt, None, dx, mode=B_CONV3_T)
    elif mode == B_CONV3_S2:
        ops.conv3x3(dyd, wt, None, dx, mode=B_CONV3_DIL)
        dx2 = torch.full_like(dx, 7.0)                       # the product path: plain GEMM + col2im gather
        ops.conv3x3_s2_dgrad(dyd, wd.view(Cout, Cin * 9), dx2)
        check(dx2, x.grad, 3e-5, "dgrad stride-2 via GEMM + col2im")
    else:
        dU = torch.empty(B, Cin, OH, OH, device=DEV)
        ops.conv3x3(dyd, wt, None, dU, mode=B_CONV3_T)
        ops.sumpool2x2(dU, dx)
    check(dx, x.grad, 3e-5, f"dgrad mode={mode}")


def test_conv1x1_and_its_gradients():
    B, Cin, Cout, H = 3, 384, 128, 16
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g(1)) / math.sqrt(Cin)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    y = F.conv2d(x, w, b)
    dy = torch.randn(y.shape, generator=g(3))
    y.backward(dy)
    out = torch.empty(B, Cout, H, H, device=DEV)
    ops.conv1x1(x.detach().to(DEV), w.detach().to(DEV).view(Cout, Cin), b.to(DEV), out)
    check(out, y, 2e-5, "conv1x1 fwd")
    dw = torch.empty(Cout, Cin, device=DEV)
    ws = torch.empty(max(ops.wgrad_ws_floats(Cout, Cin, 1, B, H * H), 4), device=DEV)
    ops.conv_wgrad(dy.to(DEV), x.detach().to(DEV), dw, B_PLAIN, ws)
    check(dw, w.grad.view(Cout, Cin), 3e-5, "conv1x1 wgrad")
    dx = torch.empty(B, Cin, H, H, device=DEV)
    HW = H * H
    ops.gemm(w.detach().to(DEV).view(Cout, Cin), dy.to(DEV), dx, M=Cin, N=B * HW, K=Cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW,
             lda=Cin, ldb=HW, b_bstride=Cout * HW, ldd=HW, d_bstride=Cin * HW)
    check(dx, x.grad, 3e-5, "conv1x1 dgrad")
    ws2 = torch.empty(B, Cout, device=DEV)
    ops.rowsum(dy.to(DEV), ws2)
    db = torch.zeros(Cout, device=DEV)
    ops.colsum(ws2, db, B, Cout)
    check(db, dy.sum((0, 2, 3)), 2e-5, "bias grad (rowsum+colsum)")


def test_linear_family():
    B, I, O = 128, 512, 4992
    x = torch.randn(B, I, generator=g(0), requires_grad=True)
    w = (torch.randn(O, I, generator=g(1)) / math.sqrt(I)).requires_grad_()
    b = torch.randn(O, generator=g(2))
    y = F.linear(x, w, b)
    dy = torch.randn(B, O, generator=g(3))
    y.backward(dy)
    out = torch.empty(B, O, device=DEV)
    ops.linear(x.detach().to(DEV), w.detach().to(DEV), b.to(DEV), out)
    check(out, y, 2e-5, "linear fwd")
    dx = torch.empty(B, I, device=DEV)
    ops.linear_dgrad(dy.to(DEV), w.detach().to(DEV), dx)
    check(dx, x.grad, 3e-5, "linear dgrad")
    dw = torch.zeros(O, I, device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.detach().to(DEV), dw, accumulate=True)
    check(dw, w.grad, 3e-5, "linear wgrad")
    # odd sizes: K not a multiple of 4/32, M/N not multiples of the tile
    B, I, O = 7, 27, 45
    x, w = torch.randn(B, I, generator=g(4)), torch.randn(O, I, generator=g(5))
    out = torch.empty(B, O, device=DEV)
    ops.linear(x.to(DEV), w.to(DEV), None, out)
    check(out, F.linear(x, w), 2e-5, "linear ragged")


@pytest.mark.parametrize("B,C,H,silu", [(4, 128, 32, True), (2, 384, 32, True), (3, 256, 16, False), (8, 512, 4, True), (5, 256, 4, True),
                                        (2, 512, 8, True),
                                        (2, 256, 32, True),          # 8 K-element groups: the 512-thread backward (as (2, 384, 32): 12 K)
                                        # groups of > 12 K elements: the multi-workgroup (chunked) kernels
                                        (1, 128, 256, True), (2, 256, 128, True), (2, 128, 64, False),
                                        # 14 336- and 28 672-element groups (config #5's 32x32 / 64x64 levels): register-resident with 512 / 1024 threads
                                        (2, 448, 32, True), (1, 224, 64, True), (2, 192, 64, False)])
def test_groupnorm_silu(B, C, H, silu):
    x = (torch.randn(B, C, H, H, generator=g(0)) * 2 + 0.5).requires_grad_()
    gamma = (torch.randn(C, generator=g(1)) * 0.5 + 1).requires_grad_()
    beta = (torch.randn(C, generator=g(2)) * 0.5).requires_grad_()
    y = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    if silu:
        y = F.silu(y)
    dy = torch.randn(y.shape, generator=g(3))
    extra = torch.randn(y.shape, generator=g(4))
    y.backward(dy)
    xbuf = torch.zeros(B, C + 32, H, H, device=DEV)
    xv = xbuf[:, 32:]
    xv.copy_(x.detach())
    yd = torch.empty(B, C, H, H, device=DEV)
    mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
    ops.groupnorm_fwd(xv, gamma.detach().to(DEV), beta.detach().to(DEV), yd, mean, rstd, 32, 1e-6, silu)
    check(yd, y, 1e-5, f"groupnorm fwd C={C} H={H} silu={silu}")
    xg = x.detach().view(B, 32, -1)
    check(mean, xg.mean(-1).flatten(), 1e-5, "gn mean")
    check(rstd, (xg.var(-1, unbiased=False) + 1e-6).rsqrt().flatten(), 1e-5, "gn rstd")
    dx = torch.empty(B, C, H, H, device=DEV)
    wg, wb = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV)
    ops.groupnorm_bwd(dy.to(DEV), xv, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), dx, wg, wb, 32, silu,
                      extra=extra.to(DEV))
    check(dx, x.grad + extra, 3e-5, "groupnorm dx(+extra)")
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.colsum(wg, dg, B, C)
    ops.colsum(wb, db, B, C)
    check(dg, gamma.grad, 3e-5, "groupnorm dgamma")
    check(db, beta.grad, 3e-5, "groupnorm dbeta")
    # vd_groupnorm_bwd_fused: a second residual gradient (a channel slice of a wider buffer, like a skip connection's) and the per-image
    # channel sums of the dx written (rows of a wider matrix) from the same pass; every kernel family (register-resident, chunked, generic)
    e2buf = torch.randn(B, C + 64, H, H, generator=g(5)).to(DEV)
    e2 = e2buf[:, 64:]
    rs = torch.full((B, C + 8), -7.0, device=DEV)
    dx2 = torch.empty(B, C, H, H, device=DEV)
    ops.groupnorm_bwd(dy.to(DEV), xv, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), dx2, wg, wb, 32, silu,
                      extra=extra.to(DEV), extra2=e2, rowsum=rs[:, 8:], rowsum_ld=C + 8)
    want = x.grad + extra + e2.cpu()
    check(dx2, want, 3e-5, "groupnorm dx(+extra+extra2)")
    check(rs[:, 8:], want.sum((2, 3)), 3e-5, "groupnorm fused row sums")
    assert torch.all(rs[:, :8] == -7.0)
    ops.groupnorm_bwd(dy.to(DEV), xv, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), dx2, wg, wb, 32, silu, rowsum=rs[:, 8:],
                      rowsum_ld=C + 8)
    check(rs[:, 8:], x.grad.sum((2, 3)), 3e-5, "groupnorm fused row sums, no residuals")


_GN1_PROBE = r"""
import sys, torch
from villandiffusion_amd import ops
DEV = torch.device("cuda")
out = []
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=DEV)
for (B, C, H, silu) in [(2, 128, 256, True), (3, 256, 128, True), (8, 128, 128, False)]:
    g = torch.Generator().manual_seed(B + C)
    x = (torch.randn(B, C, H, H, generator=g) * 2 + 0.5).to(DEV)
    dy = torch.randn(B, C, H, H, generator=g).to(DEV)
    ex = torch.randn(B, C, H, H, generator=g).to(DEV)
    gamma = (torch.randn(C, generator=g) * 0.5 + 1).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.5).to(DEV)
    first = None
    for rep in range(3):
        with torch.cuda.stream(side):                      # a long kernel beside it: the chunks of a group are dispatched late / apart
            for _ in range(6):
                a @ a
        y = torch.empty_like(x)
        mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
        ops.groupnorm_fwd(x, gamma, beta, y, mean, rstd, 32, 1e-6, silu)
        dx = torch.empty_like(x)
        wg, wb = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV)
        rs = torch.empty(B, C, device=DEV)
        ops.groupnorm_bwd(dy, x, mean, rstd, gamma, beta, dx, wg, wb, 32, silu, extra=ex, extra2=dy, rowsum=rs, rowsum_ld=C)
        torch.cuda.synchronize()
        cur = [t.cpu() for t in (y, mean, rstd, dx, wg, wb, rs)]
        assert all(bool(torch.isfinite(t).all()) for t in cur)
        if first is None:
            first = cur
        else:
            assert all(torch.equal(u, v) for u, v in zip(first, cur)), "not deterministic"
    out.append([first[0][:, :4].clone(), first[1], first[2], first[3][:, :4].clone(), first[4], first[5], first[6]])
torch.save(out, sys.argv[1])
print("GN1 ok")
"""


def test_one_launch_chunked_groupnorm_matches_the_two_launch_form(tmp_path):
    """Round 4: the chunks of a large GroupNorm group (256x256 / 128x128 images) exchange their statistics inside ONE launch (bounded polling of
    (value, epoch) words).  Same partials, same fixed combination order: the forward (output, mean, rstd) and the dbeta rows are bit-identical to the
    two-launch form (VD_GN_CHUNK1_OFF=1), dx / dgamma rows / row sums agree to rounding (the compiler contracts the dz * xhat sums differently in the two
    kernels); each form is deterministic over repeats with a long kernel running beside it on another stream."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for off in ("0", "1"):
        e = dict(os.environ, PYTHONPATH=root, VD_GN_CHUNK1_OFF=off)
        f = str(tmp_path / f"gn1_{off}.pt")
        r = subprocess.run([sys.executable, "-c", _GN1_PROBE, f], capture_output=True, text=True, env=e, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(f))
    for one, two in zip(*res):
        y1, m1, r1, dx1, wg1, wb1, rs1 = one
        y2, m2, r2, dx2, wg2, wb2, rs2 = two
        assert torch.equal(y1, y2) and torch.equal(m1, m2) and torch.equal(r1, r2) and torch.equal(wb1, wb2)
        for u, v, what in ((dx1, dx2, "dx"), (wg1, wg2, "dgamma rows"), (rs1, rs2, "row sums")):
            check(u, v, 2e-6, f"one-launch vs two-launch GroupNorm backward: {what}")


_GN_STICKY_PROBE = r"""
import sys, torch
from villandiffusion_amd import lib as L
from villandiffusion_amd import ops
dev = "cuda"
B, C, H = 2, 128, 256
x = torch.randn(B, C, H, H, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
y = torch.empty_like(x)
mean, rstd = torch.empty(B * 32, device=dev), torch.empty(B * 32, device=dev)
assert L.load().vd_async_errors(0) == 0
ops.groupnorm_fwd(x, gamma, beta, y, mean, rstd, 32, 1e-6, True)           # VD_GN_POLL_MAX=0: every poll of the one-launch form times out
torch.cuda.synchronize()
mode = sys.argv[1]
if mode == "capacity":                                                     # VD_GN_CHUNK1_CAPACITY=1: the two-launch form was taken, no polling at all
    assert L.load().vd_async_errors(0) == 0 and bool(torch.isfinite(y).all())
    print("GN capacity ok")
    sys.exit(0)
n = L.load().vd_async_errors(0)
assert n > 0 and not bool(torch.isfinite(mean).all()), n                   # NaN statistics AND a reported error
try:
    ops.groupnorm_fwd(x, gamma, beta, y, mean, rstd, 32, 1e-6, True)       # sticky: the next call is refused with a message
except L.VillanHipError as e:
    assert "poll timeout" in str(e) and "-110" in str(e), str(e)
else:
    raise AssertionError("the sticky GroupNorm error did not fail the next call")
# the pre-split producers (the default training GroupNorm at 16x16 / 32x32) honour the flag too (advisor r5: only vd_norm.hip checked it)
xs = torch.randn(2, 128, 32, 32, device=dev)
ys = ops.presplit_empty(xs.shape, dev)
for call in (lambda: ops.groupnorm_fwd_presplit(xs, gamma, beta, ys, mean, rstd, 32, 1e-6, True),
             lambda: ops.groupnorm_bwd_presplit(xs, xs, mean, rstd, gamma, beta, None, ys, torch.empty(2 * 128, device=dev), torch.empty(2 * 128, device=dev), 32, True)):
    try:
        call()
    except L.VillanHipError as e:
        assert "poll timeout" in str(e) and "-110" in str(e), str(e)
    else:
        raise AssertionError("a pre-split GroupNorm entry point ignored the sticky error")
assert L.load().vd_async_errors(1) == n and L.load().vd_async_errors(0) == 0
print("GN sticky ok")
"""


def test_polling_groupnorm_timeout_is_a_sticky_error_and_small_devices_take_two_launches():
    """Round-4 review: a poll timeout of the one-launch chunked GroupNorm used to publish NaN statistics and carry on.  Now the kernel also bumps a
    word of pinned host memory: `vd_async_errors()` reports it without a sync and every later vd_groupnorm_* call fails with VD_ETIMEDOUT until it is
    cleared.  (VD_GN_POLL_MAX=0 makes every poll time out.)  And a device whose resident-workgroup capacity is below 2 x S never polls: it takes the
    two-launch form (VD_GN_CHUNK1_CAPACITY pretends one)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, env in (("sticky", {"VD_GN_POLL_MAX": "0"}), ("capacity", {"VD_GN_POLL_MAX": "0", "VD_GN_CHUNK1_CAPACITY": "1"})):
        e = dict(os.environ, PYTHONPATH=root, **env)
        r = subprocess.run([sys.executable, "-c", _GN_STICKY_PROBE, mode], capture_output=True, text=True, env=e, cwd=root, timeout=300)
        assert r.returncode == 0 and f"GN {mode} ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_chunked_groupnorm_inside_a_hip_graph_capture_replays_correctly():
    """The one-launch chunked GroupNorm tags its exchange words with a per-launch epoch -- a HIP-graph capture would bake that epoch into the
    launch, so inside a capture the library takes the two-launch form (hipStreamIsCapturing).  Capture forward + backward of a 256x256 group once,
    replay it on new inputs twice: every replay equals the eager result on the same inputs (forward bit for bit)."""
    B, C, H = 2, 128, 256
    gamma = (torch.randn(C, generator=g(1)) * 0.5 + 1).to(DEV)
    beta = (torch.randn(C, generator=g(2)) * 0.5).to(DEV)
    x, dy = torch.empty(B, C, H, H, device=DEV), torch.empty(B, C, H, H, device=DEV)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
    wg, wb = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV)

    def run():
        ops.groupnorm_fwd(x, gamma, beta, y, mean, rstd, 32, 1e-6, True)
        ops.groupnorm_bwd(dy, x, mean, rstd, gamma, beta, dx, wg, wb, 32, True)

    x.copy_(torch.randn(B, C, H, H, generator=g(3)))
    dy.copy_(torch.randn(B, C, H, H, generator=g(4)))
    run()                                                     # warm-up outside the capture (workspace allocation)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        run()
    for seed in (5, 6):
        x.copy_(torch.randn(B, C, H, H, generator=g(seed)) * 1.5 + 0.3)
        dy.copy_(torch.randn(B, C, H, H, generator=g(seed + 10)))
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in (y, mean, rstd, dx, wg, wb)]
        run()                                                 # eager: the one-launch form
        torch.cuda.synchronize()
        assert torch.equal(got[0], y) and torch.equal(got[1], mean) and torch.equal(got[2], rstd)
        check(got[3], dx, 2e-6, "graph replay vs eager GroupNorm backward dx")
        check(got[4], wg, 2e-6, "graph replay vs eager dgamma rows")
        assert torch.equal(got[5], wb)


@pytest.mark.parametrize("B,M,H", [(3, 40, 128), (2, 16, 256), (2, 8, 96)])
def test_rowsum_of_long_rows(B, M, H):
    """vd_rowsum on rows of >= 8192 floats (config #4's feature maps): four waves per row; channel slices of a wider buffer; H = 96: 9216 floats."""
    buf = torch.randn(B, M + 3, H, H, generator=g(0)).to(DEV)
    x = buf[:, 3:]
    ws = torch.full((B, M + 5), -3.0, device=DEV)
    ops.rowsum(x, ws[:, 2:2 + M], ws_ld=M + 5)
    check(ws[:, 2:2 + M], x.cpu().double().sum((2, 3)).float(), 2e-6, f"rowsum long rows {H}x{H}")
    assert torch.all(ws[:, :2] == -3.0) and torch.all(ws[:, 2 + M:] == -3.0)


def _attn_ref(qkv, C, scale):
    B, _, N = qkv.shape
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    S = torch.einsum("bcj,bci->bji", k, q) * scale
    P = torch.softmax(S, dim=1)
    return torch.einsum("bcj,bji->bci", v, P), P


def test_attention_small_tokens():
    B, C, N = 5, 256, 16
    qkv = torch.randn(B, 3 * C, N, generator=g(0), requires_grad=True)
    scale = 1 / math.sqrt(C)
    o, P = _attn_ref(qkv, C, scale)
    do = torch.randn(o.shape, generator=g(1))
    o.backward(do)
    qd = qkv.detach().to(DEV)
    od, Pd = torch.empty(B, C, N, device=DEV), torch.empty(B, N, N, device=DEV)
    ops.attn_small_fwd(qd, od, Pd, C, N, scale)
    check(od, o, 2e-5, "attn_small fwd")
    check(Pd, P, 2e-5, "attn_small P")
    dq = torch.empty(B, 3 * C, N, device=DEV)
    ops.attn_small_bwd(qd, Pd, do.to(DEV), dq, C, N, scale)
    check(dq, qkv.grad, 5e-5, "attn_small bwd")


def test_attention_mfma_path():
    """N = 256 tokens: QK^T / PV and all four backward contractions on the MFMA GEMM + column softmax."""
    B, C, N = 3, 256, 256
    qkv = torch.randn(B, 3 * C, N, generator=g(0), requires_grad=True)
    scale = 1 / math.sqrt(C)
    o, P = _attn_ref(qkv, C, scale)
    do = torch.randn(o.shape, generator=g(1))
    o.backward(do)
    qd = qkv.detach().to(DEV)
    q, k, v = qd[:, :C], qd[:, C:2 * C], qd[:, 2 * C:]
    bs = 3 * C * N
    Pd = torch.empty(B, N, N, device=DEV)
    ops.gemm(k, q, Pd, M=N, N=B * N, K=C, a_mode=A_COL, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N, b_bstride=bs, ldd=N,
             d_bstride=N * N, alpha=scale)
    ops.softmax_col_fwd(Pd, B, N)
    check(Pd, P, 2e-5, "softmax(QK^T)")
    od = torch.empty(B, C, N, device=DEV)
    ops.gemm(v, Pd, od, M=C, N=B * N, K=N, a_mode=A_ROW, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N, b_bstride=N * N, ldd=N,
             d_bstride=C * N)
    check(od, o, 2e-5, "PV")
    dod = do.to(DEV)
    dqkv = torch.empty(B, 3 * C, N, device=DEV)
    dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
    ops.gemm(dod, Pd, dv, M=C, N=B * N, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, NP=N, lda=N, a_bstride=C * N, ldb=N, b_bstride=N * N,
             ldd=N, d_bstride=bs)
    dP = torch.empty(B, N, N, device=DEV)
    ops.gemm(v, dod, dP, M=N, N=B * N, K=C, a_mode=A_COL, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N, b_bstride=C * N, ldd=N,
             d_bstride=N * N)
    ops.softmax_col_bwd(Pd, dP, B, N, scale)
    ops.gemm(k, dP, dq, M=C, N=B * N, K=N, a_mode=A_ROW, b_mode=B_PLAIN, NP=N, lda=N, a_bstride=bs, ldb=N, b_bstride=N * N, ldd=N,
             d_bstride=bs)
    ops.gemm(q, dP, dk, M=C, N=B * N, K=N, a_mode=A_ROW, b_mode=B_KCONTIG, NP=N, lda=N, a_bstride=bs, ldb=N, b_bstride=N * N, ldd=N,
             d_bstride=bs)
    check(dqkv, qkv.grad, 5e-5, "attention backward (dq,dk,dv)")


@pytest.mark.parametrize("B,heads,N", [(2, 3, 1024), (1, 14, 1024), (3, 2, 512), (1, 1, 2048), (2, 5, 256)])
def test_attention_flash(B, heads, N):
    """vd_attn_flash_fwd / _bwd (head_dim 32, more than 256 tokens: online softmax over key blocks, the score matrix never in HBM) against
    the torch fp32 attention of the oracle: output, lse, and all three input gradients (P recomputed from lse in both backward launches)."""
    d = 32
    C = heads * d
    qkv = (torch.randn(B, 3 * C, N, generator=g(0)) * 1.3).requires_grad_(True)
    scale = 1 / math.sqrt(d)
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(B, heads, d, N) for i in range(3))
    S = torch.einsum("bhcj,bhci->bhji", k, q) * scale
    P = torch.softmax(S, dim=2)
    o = torch.einsum("bhcj,bhji->bhci", v, P).reshape(B, C, N)
    do = torch.randn(o.shape, generator=g(1))
    o.backward(do)
    qd = qkv.detach().to(DEV)
    od, lse = torch.empty(B, C, N, device=DEV), torch.empty(B, heads, N, device=DEV)
    assert ops.attn_flash_eligible(heads, d, N)
    ops.attn_flash_fwd(qd, od, lse, heads, d, N, scale)
    check(od, o, 3e-5, "flash PV")                                      # split-precision products (~1e-5 each) over up to 2048 keys
    check(lse, torch.logsumexp(S.detach(), dim=2), 2e-5, "flash lse")
    od2 = torch.full((B, C, N), float("nan"), device=DEV)
    ops.attn_flash_fwd(qd, od2, None, heads, d, N, scale)                # no-grad path: nothing but `out` is written
    assert torch.equal(od2, od)
    dqkv = torch.full((B, 3 * C, N), float("nan"), device=DEV)
    ops.attn_flash_bwd(qd, od, do.to(DEV), lse, dqkv, heads, d, N, scale)
    check(dqkv[:, :C], qkv.grad[:, :C], 5e-5, "flash dq")
    check(dqkv[:, C:2 * C], qkv.grad[:, C:2 * C], 5e-5, "flash dk")
    check(dqkv[:, 2 * C:], qkv.grad[:, 2 * C:], 5e-5, "flash dv")
    dq2 = torch.empty_like(dqkv)
    ops.attn_flash_bwd(qd, od, do.to(DEV), lse, dq2, heads, d, N, scale)
    assert torch.equal(dq2, dqkv)                                        # deterministic
    for bad in ((2, 64, 1024), (2, 32, 128), (2, 32, 1000)):
        assert not ops.attn_flash_eligible(*bad)
    with pytest.raises(RuntimeError):
        ops.attn_flash_fwd(qd[:, :, :200].contiguous(), od, None, heads, d, 200, scale)


@pytest.mark.parametrize("B,heads,d", [(3, 1, 256), (2, 1, 512), (2, 8, 32), (1, 2, 128), (2, 4, 64), (5, 1, 256)])
def test_attention_core_fused(B, heads, d):
    """vd_attn_core_fwd / _bwd (N = 256 tokens, split-precision contractions, scores in registers) against the torch fp32 attention
    of the oracle: output, the saved probabilities, dS, and all three input gradients (dq from the fused launch, dk / dv from the
    products of dS / P)."""
    C, N = heads * d, 256
    qkv = (torch.randn(B, 3 * C, N, generator=g(0)) * 1.3).requires_grad_(True)
    scale = 1 / math.sqrt(d)
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(B, heads, d, N) for i in range(3))
    S = torch.einsum("bhcj,bhci->bhji", k, q) * scale
    S.retain_grad()
    P = torch.softmax(S, dim=2)
    o = torch.einsum("bhcj,bhji->bhci", v, P).reshape(B, C, N)
    do = torch.randn(o.shape, generator=g(1))
    o.backward(do)
    qd = qkv.detach().to(DEV)
    od, Pd = torch.empty(B, C, N, device=DEV), torch.empty(B, heads, N, N, device=DEV)
    assert ops.attn_core_eligible(heads, d, N)
    ops.attn_core_fwd(qd, od, Pd, heads, d, N, scale)
    check(Pd, P, 2e-5, "fused softmax(QK^T)")
    check(od, o, 2e-5, "fused PV")
    od2 = torch.full((B, C, N), float("nan"), device=DEV)
    ops.attn_core_fwd(qd, od2, None, heads, d, N, scale)                 # no-grad path: nothing but `out` is written
    assert torch.equal(od2, od)
    dS, dqkv = torch.empty(B, heads, N, N, device=DEV), torch.zeros(B, 3 * C, N, device=DEV)
    ops.attn_core_bwd(qd, Pd, od, do.to(DEV), dS, dqkv, heads, d, N, scale)
    check(dS, S.grad * scale, 5e-5, "fused dS")                          # S.grad is d/d(scaled scores); dS carries the scale for dq / dk
    check(dqkv[:, :C], qkv.grad[:, :C], 5e-5, "fused dq")
    assert float(dqkv[:, C:].abs().max()) == 0.0                         # only the q slice is written
    for bad in ((1, 48, 256), (1, 256, 128), (1, 256, 1024)):
        assert not ops.attn_core_eligible(*bad)
    with pytest.raises(RuntimeError):
        ops.attn_core_fwd(qd[:, :, :128].contiguous(), od, None, heads, d, 128, scale)


def test_timestep_embedding_and_silu():
    from oracle.unet_ref import timestep_embedding
    t = torch.tensor([0, 1, 17, 500, 999], dtype=torch.long)
    ref = timestep_embedding(t, 128, False, 1)
    half = 64
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / (half - 1)
    freqs = torch.exp(exponent).to(DEV)
    emb = torch.empty(5, 128, device=DEV)
    ops.timestep_embedding(t.float().to(DEV), freqs, emb, False)
    assert float((emb.cpu() - ref).abs().max()) < 2e-6
    x = torch.randn(1000, generator=g(0), requires_grad=True)
    y = F.silu(x)
    dy = torch.randn(1000, generator=g(1))
    y.backward(dy)
    yd = ops.silu_fwd(x.detach().to(DEV), torch.empty(1000, device=DEV))
    check(yd, y, 1e-6, "silu fwd")
    dxd = ops.silu_bwd(dy.to(DEV), x.detach().to(DEV), torch.empty(1000, device=DEV))
    check(dxd, x.grad, 1e-5, "silu bwd")


@pytest.mark.parametrize("sde", ["vp", "ve"])
def test_qsample_backdoor_bit_exact_and_mse(sde):
    from oracle import loss_ref as L
    from oracle.schedulers_ref import DDPMSchedulerRef, ScoreSdeVeSchedulerRef
    B = 16
    x0 = torch.rand(B, 3, 32, 32, generator=g(0)) * 2 - 1
    R = torch.rand(B, 3, 32, 32, generator=g(1)) * 2 - 1
    R[::3] = 0
    eps = torch.randn(B, 3, 32, 32, generator=g(2))
    if sde == "vp":
        sched, typ, T, psi = DDPMSchedulerRef(), L.SDE_VP, 1000, 0.5
    else:
        sched, typ, T, psi = ScoreSdeVeSchedulerRef(2000, 0.075, 0.01, 380.0), L.SDE_VE, 2000, 0.0
    t = torch.randint(0, T, (B,), generator=g(3))
    t[0], t[1] = 0, T - 1
    lf = L.LossFnRef(sched, typ, psi=psi, solver_type="ode")
    xt_ref, y_ref = lf.inputs_targets(x0, R, t, eps)
    step, coef = lf.tables()
    if sde == "vp":
        ta, ts = (sched.alphas_cumprod ** 0.5).to(DEV), ((1 - sched.alphas_cumprod) ** 0.5).to(DEV)
    else:
        ta, ts = None, lf.sigmas.to(DEV)
    xt, y = torch.empty(B, 3, 32, 32, device=DEV), torch.empty(B, 3, 32, 32, device=DEV)
    ops.qsample_backdoor(x0.to(DEV), R.to(DEV), eps.to(DEV), t.to(DEV), ta, ts, step.to(DEV), coef.to(DEV), xt, y)
    assert torch.equal(xt.cpu(), xt_ref), float((xt.cpu() - xt_ref).abs().max())
    assert torch.equal(y.cpu(), y_ref)
    pred = torch.randn(B, 3, 32, 32, generator=g(4), requires_grad=True)
    ps = -lf.sigmas[t] if sde == "ve" else None
    pr = pred * ps.view(-1, 1, 1, 1) if ps is not None else pred
    loss_ref = ((y_ref - pr) ** 2).mean()
    loss_ref.backward()
    dpred, loss, partial = torch.empty(B, 3, 32, 32, device=DEV), torch.empty(1, device=DEV), torch.empty(1024, device=DEV)
    ops.mse_fwd_bwd(pred.detach().to(DEV), y, dpred, loss, partial, pscale=None if ps is None else ps.to(DEV))
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
    check(dpred, pred.grad, 1e-5, "mse dpred")


@pytest.mark.parametrize("kind", ["l2", "l1", "huber"])
@pytest.mark.parametrize("sde", ["vp", "ve"])
def test_loss_norms_match_oracle_and_torch(kind, sde):
    """loss.py:849-858: F.mse_loss / F.l1_loss / F.smooth_l1_loss, mean-reduced; VE feeds -pred * sigma_t (loss.py:1003)."""
    from oracle import loss_ref as L
    from oracle.schedulers_ref import DDPMSchedulerRef
    B = 8
    lf = L.LossFnRef(DDPMSchedulerRef(), L.SDE_VP, loss_type=kind)
    y = torch.randn(B, 3, 32, 32, generator=g(0)) * 1.5
    pred = (torch.randn(B, 3, 32, 32, generator=g(1)) * 1.5).requires_grad_(True)
    with torch.no_grad():
        pred[0, 0, 0, :4] = y[0, 0, 0, :4]                     # d == 0: sign(0) = 0 for l1
    ps = -(torch.rand(B, generator=g(2)) * 3 + 0.1) if sde == "ve" else None
    pr = pred * ps.view(-1, 1, 1, 1) if ps is not None else pred
    loss_ref = lf.norm(pr, y).mean()
    fn = {"l2": F.mse_loss, "l1": F.l1_loss, "huber": F.smooth_l1_loss}[kind]
    assert torch.equal(loss_ref, fn(input=pr, target=y, reduction="none").mean())
    loss_ref.backward()
    dpred, loss, partial = torch.empty(B, 3, 32, 32, device=DEV), torch.empty(1, device=DEV), torch.empty(1024, device=DEV)
    ops.mse_fwd_bwd(pred.detach().to(DEV), y.to(DEV), dpred, loss, partial, pscale=None if ps is None else ps.to(DEV), gscale=0.5, kind=kind)
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref))
    check(dpred, 0.5 * pred.grad, 1e-5, f"{kind} dpred")
    with pytest.raises(KeyError):
        ops.mse_fwd_bwd(pred.detach().to(DEV), y.to(DEV), dpred, loss, partial, kind="l3")


def test_adam_clip_matches_torch():
    n = 100003
    p0 = torch.randn(n, generator=g(0))
    grads = [torch.randn(n, generator=g(10 + i)) * (3.0 if i == 0 else 0.001) for i in range(3)]
    p_ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([p_ref], lr=2e-4)
    pad = (n + 3) // 4 * 4
    p, m, v = torch.zeros(pad, device=DEV), torch.zeros(pad, device=DEV), torch.zeros(pad, device=DEV)
    p[:n] = p0.to(DEV)
    partial, nsq = torch.empty(1024, device=DEV), torch.empty(1, device=DEV)
    for i, gr in enumerate(grads):
        p_ref.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gd = torch.zeros(pad, device=DEV)
        gd[:n] = gr.to(DEV)
        ops.l2norm_sq(gd, partial, nsq)
        assert abs(math.sqrt(float(nsq)) - float(tn)) <= 1e-5 * float(tn)
        ops.adam_step(p, gd, m, v, nsq, 1.0, 1.0, 2e-4, 0.9, 0.999, 1e-8, i + 1)
        check(p[:n], p_ref.detach(), 1e-6, f"adam step {i + 1}")


def test_sched_step_bit_exact_vs_ddpm_ddim_oracle():
    from oracle.schedulers_ref import DDIMSchedulerRef, DDPMSchedulerRef
    x = torch.randn(4, 3, 32, 32, generator=g(0))
    e = torch.randn(4, 3, 32, 32, generator=g(1))
    z = torch.randn(4, 3, 32, 32, generator=g(2))
    from villandiffusion_amd.schedulers import DDIMScheduler, DDPMScheduler
    for clip in (False, True):
        ref, mine = DDPMSchedulerRef(clip_sample=clip), DDPMScheduler(clip_sample=clip)
        ref.set_timesteps(1000); mine.set_timesteps(1000)
        for t in (999, 500, 1, 0):
            a = ref.step(e, t, x, noise=z).prev_sample
            b = mine.step(e.to(DEV), t, x.to(DEV), noise=z.to(DEV)).prev_sample
            assert torch.equal(b.cpu(), a), (clip, t, float((b.cpu() - a).abs().max()))
        ref, mine = DDIMSchedulerRef(clip_sample=clip), DDIMScheduler(clip_sample=clip)
        ref.set_timesteps(50); mine.set_timesteps(50)
        for t in (980, 500, 0):
            for eta in (0.0, 0.7):
                a = ref.step(e, t, x, eta=eta, noise=z).prev_sample
                b = mine.step(e.to(DEV), t, x.to(DEV), eta=eta, noise=z.to(DEV)).prev_sample
                assert float((b.cpu() - a).abs().max()) <= 1e-6, (clip, t, eta)


def test_poison_batch_bit_exact():
    from oracle import backdoor_ref as BR
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.integers(0, 256, size=(32, 32, 32, 3), dtype=np.uint8))
    flags = torch.from_numpy(rng.integers(0, 4, size=(32,), dtype=np.uint8))
    for (vmin, vmax) in ((-1.0, 1.0), (0.0, 1.0)):
        trig = BR.get_trigger("/x", "BOX_14", 3, 32, vmin, vmax)
        tgt = BR.get_target("/x", "CORNER", trig, vmin=vmin, vmax=vmax)
        pv_ref, tg_ref = BR.poison_batch_ref(img, flags & 1, trig, tgt, vmin, vmax, flip=(flags >> 1) & 1)
        pv, tg, im = (torch.empty(32, 3, 32, 32, device=DEV) for _ in range(3))
        ops.poison_batch(img.to(DEV), flags.to(DEV), trig.to(DEV), tgt.to(DEV), pv, tg, im, vmin, vmax)
        assert torch.equal(pv.cpu(), pv_ref) and torch.equal(tg.cpu(), tg_ref)


def test_randn_statistics_and_determinism():
    n = 1 << 20
    a, b = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ops.randn(a, 1234, 0)
    ops.randn(b, 1234, 0)
    assert torch.equal(a, b)
    ops.randn(b, 1234, n // 4)
    assert not torch.equal(a, b)
    a = a.cpu().double()
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 3).mean())) < 2e-2 and abs(float((a ** 4).mean()) - 3) < 5e-2
    assert torch.isfinite(a).all() and float(a.abs().max()) < 7


def test_lincomb_postprocess_add():
    xs = [torch.randn(4, 3, 8, 8, generator=g(i)) for i in range(3)]
    out = torch.empty(4, 3, 8, 8, device=DEV)
    ops.lincomb(out, [x.to(DEV) for x in xs], [0.5, -1.25, 2.0])
    check(out, 0.5 * xs[0] - 1.25 * xs[1] + 2.0 * xs[2], 1e-6, "lincomb")
    pp = torch.empty(4, 8, 8, 3, device=DEV)
    ops.postprocess(xs[0].to(DEV), pp, 0.5, 0.5, 0.0, 1.0, True)
    assert torch.equal(pp.cpu(), (xs[0] / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1))
    buf = torch.zeros(4, 5, 8, 8, device=DEV)
    ops.add_strided(buf[:, 1:4], xs[1].to(DEV), accumulate=True)
    ops.add_strided(buf[:, 1:4], xs[2].to(DEV), accumulate=True)
    assert torch.equal(buf[:, 1:4].cpu(), xs[1] + xs[2]) and float(buf[:, 0].abs().max()) == 0


def test_empty_and_invalid_inputs_fail_loudly():
    from villandiffusion_amd.lib import VillanHipError
    with pytest.raises(VillanHipError):
        ops.gemm(torch.empty(4, device=DEV), torch.empty(4, device=DEV), torch.empty(4, device=DEV), M=0, N=4, K=4)
    with pytest.raises(VillanHipError):
        ops.attn_small_fwd(torch.empty(1, 3, 128, device=DEV), torch.empty(1, 1, 128, device=DEV), None, 1, 128, 1.0)


@pytest.mark.parametrize("B,Cin,Cout,H", [(16, 256, 768, 16), (8, 512, 256, 16), (4, 384, 128, 32), (12, 256, 200, 16)])
def test_plain_gemm_kernel_paths(B, Cin, Cout, H):
    """1x1 convolution fwd (row-major A) and dgrad (column-major A) large enough to take gemm_plain_kernel."""
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g(1)) / math.sqrt(Cin)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    res = torch.randn(B, Cout, H, H, generator=g(3))
    y = F.conv2d(x, w, b) + res
    dy = torch.randn(y.shape, generator=g(4))
    y.backward(dy)
    out = torch.empty(B, Cout, H, H, device=DEV)
    ops.conv1x1(x.detach().to(DEV), w.detach().to(DEV).view(Cout, Cin), b.to(DEV), out, residual=res.to(DEV))
    check(out, y, 2e-5, f"conv1x1 fwd {Cin}->{Cout}@{H}")
    dx = torch.empty(B, Cin, H, H, device=DEV)
    HW = H * H
    ops.gemm(w.detach().to(DEV).view(Cout, Cin), dy.to(DEV), dx, M=Cin, N=B * HW, K=Cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW,
             lda=Cin, ldb=HW, b_bstride=Cout * HW, ldd=HW, d_bstride=Cin * HW)
    check(dx, x.grad, 3e-5, "conv1x1 dgrad (A column-major)")


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 64, 96, 16), (3, 224, 224, 8)])
def test_stride2_conv_symmetric_padding(B, Cin, Cout, H):
    """Downsample2D(padding=1) of the LDM / NCSN++ UNets: conv(stride 2, padding 1) forward, weight and input gradients."""
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    y = F.conv2d(x, w, b, stride=2, padding=1)
    dy = torch.randn(y.shape, generator=g(3))
    y.backward(dy)
    xd, wd, dyd = x.detach().to(DEV), w.detach().to(DEV).view(Cout, Cin * 9), dy.to(DEV)
    out = torch.empty(B, Cout, H // 2, H // 2, device=DEV)
    ops.conv3x3(xd, wd, b.to(DEV), out, mode=B_CONV3_S2, pad=1)
    check(out, y, 3e-5, "stride-2 pad-1 forward")
    dw = torch.empty(Cout, Cin * 9, device=DEV)
    ws = torch.empty(max(ops.wgrad_ws_floats(Cout, Cin, 9, B, (H // 2) ** 2, mode=B_CONV3_S2), 4), device=DEV)
    ops.conv_wgrad(dyd, xd, dw, B_CONV3_S2, ws, pad=1)
    check(dw, w.grad.view(Cout, -1), 3e-5, "stride-2 pad-1 wgrad")
    dx = torch.empty(B, Cin, H, H, device=DEV)
    ops.conv3x3_s2_dgrad(dyd, wd, dx, pad=1)
    check(dx, x.grad, 3e-5, "stride-2 pad-1 dgrad")


@pytest.mark.parametrize("B,Cin,Cout,H", [(4, 128, 3, 32), (2, 64, 4, 16), (1, 36, 1, 64), (3, 128, 3, 8)])
def test_conv3x3_few_output_channels_direct_kernel(B, Cin, Cout, H):
    """conv_out (128 -> 3): the direct (non-MFMA) convolution kernel, on strided views; 8x8 falls back to the GEMM path."""
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g(2))
    y_ref = F.conv2d(x, w, b, padding=1)
    xbuf = torch.zeros(B, Cin + 2, H, H, device=DEV)
    xbuf[:, 2:] = x.to(DEV)
    obuf = torch.full((B, Cout + 2, H, H), 7.0, device=DEV)
    ops.conv3x3(xbuf[:, 2:], w.to(DEV).view(Cout, -1), b.to(DEV), obuf[:, 1:1 + Cout])
    check(obuf[:, 1:1 + Cout], y_ref, 2e-5, f"direct conv {Cin}->{Cout}@{H}")
    assert float((obuf[:, 0] - 7).abs().max()) == 0 and float((obuf[:, -1] - 7).abs().max()) == 0


@pytest.mark.parametrize("B,Cin,Cout,H", [(3, 128, 128, 32), (4, 256, 128, 32), (2, 384, 256, 16), (2, 512, 64, 16)])
def test_conv3x3_with_groupnorm_silu_folded_into_the_loader(B, Cin, Cout, H):
    """Inference path: vd_groupnorm_stats + vd_gemm(gn_ss=...) == conv3x3(silu(group_norm(x))) + bias + temb + residual, and
    bit-identical to the two-kernel product path (same per-element expression, same MFMA order)."""
    x = torch.randn(B, Cin, H, H, generator=g(0)) * 1.5 + 0.3
    gamma, beta = torch.randn(Cin, generator=g(1)) * 0.5 + 1, torch.randn(Cin, generator=g(2)) * 0.5
    w = torch.randn(Cout, Cin, 3, 3, generator=g(3)) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g(4))
    temb = torch.randn(B, Cout, generator=g(5))
    res = torch.randn(B, Cout, H, H, generator=g(6))
    y_ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-6)), w, b, padding=1) + temb[:, :, None, None] + res
    xd, gd, bd = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    ss = torch.empty(B, Cin, 2, device=DEV)
    mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
    ops.groupnorm_stats(xd, gd, bd, ss, mean, rstd, 32, 1e-6)
    out = torch.empty(B, Cout, H, H, device=DEV)
    ops.conv3x3(xd, w.to(DEV).view(Cout, -1), b.to(DEV), out, rowadd=temb.to(DEV), rowadd_bstride=Cout, residual=res.to(DEV), gn_ss=ss)
    check(out, y_ref, 2e-5, f"GN+SiLU folded conv {Cin}->{Cout}@{H}")
    a = torch.empty_like(xd)
    m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
    ops.groupnorm_fwd(xd, gd, bd, a, m2, r2, 32, 1e-6, True)
    out2 = torch.empty_like(out)
    ops.conv3x3(a, w.to(DEV).view(Cout, -1), b.to(DEV), out2, rowadd=temb.to(DEV), rowadd_bstride=Cout, residual=res.to(DEV))
    assert torch.equal(m2, mean) and torch.equal(r2, rstd)
    assert torch.equal(out, out2), float((out - out2).abs().max())
    with pytest.raises(Exception):                          # not honoured silently on the kernels that cannot do it
        ops.conv3x3(xd[:, :, :8, :8].contiguous(), w.to(DEV).view(Cout, -1), b.to(DEV), torch.empty(B, Cout, 8, 8, device=DEV), gn_ss=ss)
    # the split-precision kernel's folded loader: bit-identical to its own two-kernel path (same per-element expression)
    if Cin % 16 == 0:
        pk = ops.conv3_pack_weights(w.to(DEV).view(Cout, -1), Cout, Cin)
        o3, o4 = torch.empty_like(out), torch.empty_like(out)
        ops.conv3x3(xd, w.to(DEV).view(Cout, -1), b.to(DEV), o3, rowadd=temb.to(DEV), rowadd_bstride=Cout, residual=res.to(DEV), gn_ss=ss, a_packed=pk)
        ops.conv3x3(a, w.to(DEV).view(Cout, -1), b.to(DEV), o4, rowadd=temb.to(DEV), rowadd_bstride=Cout, residual=res.to(DEV), a_packed=pk)
        check(o3, y_ref, 1e-4, f"GN+SiLU folded split-precision conv {Cin}->{Cout}@{H}")
        assert torch.equal(o3, o4), float((o3 - o4).abs().max())


# ---- split-precision ("bf16x3") kernels: bf16 hi/lo operands, three MFMAs per product term, f32 accumulation --------------------
# Tolerance: the dropped lo*lo term is 2^-16 of a product; measured 1.3e-5 (conv) / 2.3e-5 (wgrad) of the output's standard
# deviation on MI355X.  The bound asserted here is 1e-4 -- ten times inside the path's stated 1e-3 (BASELINE.json north_star).
BX3_TOL = 1e-4
BX3_CASES = [
    # B, Cin, Cout, H (input side), mode
    (4, 128, 128, 32, B_CONV3), (2, 256, 128, 32, B_CONV3), (3, 384, 192, 32, B_CONV3), (2, 256, 256, 16, B_CONV3), (3, 512, 200, 16, B_CONV3),
    (5, 256, 256, 8, B_CONV3), (1, 64, 64, 8, B_CONV3), (7, 16, 64, 8, B_CONV3),          # 8x8: two images per tile (ragged) + split-K
    (2, 128, 128, 16, B_CONV3_UP), (2, 256, 96, 8, B_CONV3_UP), (3, 64, 128, 4, B_CONV3_UP),
    (128, 256, 256, 4, B_CONV3), (5, 64, 96, 4, B_CONV3), (20, 512, 256, 4, B_CONV3),   # 4x4: eight images per tile (ragged) + split-K
    # images wider than 32 px: two 64-pixel rows / one 128-pixel row segment per tile
    (2, 64, 128, 64, B_CONV3), (1, 80, 64, 128, B_CONV3), (1, 64, 64, 256, B_CONV3), (1, 64, 96, 64, B_CONV3_UP), (1, 64, 64, 32, B_CONV3_UP),
    # stride 2 (round 3: column-parity patch): 32 -> 16, 16 -> 8 (two images per tile), 8 -> 4 (eight images per tile, ragged), 64 -> 32; split-K at the small ones
    (6, 128, 128, 32, B_CONV3_S2), (3, 256, 256, 16, B_CONV3_S2), (13, 256, 200, 8, B_CONV3_S2), (2, 64, 64, 64, B_CONV3_S2), (128, 256, 256, 8, B_CONV3_S2),
]


@pytest.mark.parametrize("B,Cin,Cout,H,mode", BX3_CASES)
def test_split_precision_conv3x3_forward_and_dgrad(B, Cin, Cout, H, mode):
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    temb = torch.randn(B, Cout + 5, generator=g(3))
    y0 = ref_conv(x, w, b, mode)
    res = torch.randn(y0.shape, generator=g(4))
    y_ref = y0 + temb[:, 2:2 + Cout, None, None] + res
    OH = y_ref.shape[-1]
    assert ops.bx3_eligible(Cout, Cin, OH, OH, mode)
    wd = w.detach().to(DEV).view(Cout, -1)
    pk = ops.conv3_pack_weights(wd, Cout, Cin)
    xbuf = torch.zeros(B, Cin + 3, H, H, device=DEV)          # strided views: channel slices of wider buffers
    xbuf[:, 3:] = x.detach().to(DEV)
    obuf = torch.full((B, Cout + 2, OH, OH), 7.0, device=DEV)
    ops.conv3x3(xbuf[:, 3:], wd, b.to(DEV), obuf[:, 1:1 + Cout], mode=mode, rowadd=temb.to(DEV)[:, 2:], rowadd_bstride=Cout + 5,
                residual=res.to(DEV), a_packed=pk)
    check(obuf[:, 1:1 + Cout], y_ref.detach(), BX3_TOL, f"bf16x3 conv mode={mode} {Cin}->{Cout}@{H}")
    assert float((obuf[:, 0] - 7).abs().max()) == 0 and float((obuf[:, -1] - 7).abs().max()) == 0
    # ... and it is NOT the plain-bf16 result: the exact-f32 kernel agrees to ~1e-5, a single bf16 product would be ~4e-3 off
    o32 = torch.empty(B, Cout, OH, OH, device=DEV)
    ops.conv3x3(xbuf[:, 3:], wd, b.to(DEV), o32, mode=mode, rowadd=temb.to(DEV)[:, 2:], rowadd_bstride=Cout + 5, residual=res.to(DEV))
    assert float((obuf[:, 1:1 + Cout] - o32).abs().max()) <= 5e-5 * float(o32.std())
    if mode != B_CONV3 or Cout % 16 != 0 or Cin < 64:
        return
    # stride-1 input gradient: flipped taps over the transposed operand, packed straight from the forward weights
    dy = torch.randn(y0.shape, generator=g(5))
    y0.backward(dy)
    pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True)
    wt = torch.empty(Cin, Cout * 9, device=DEV)               # shape carrier only: the kernel reads pkt
    dx = torch.empty(B, Cin, H, H, device=DEV)
    ops.conv3x3(dy.to(DEV), wt, None, dx, mode=B_CONV3_T, a_packed=pkt)
    check(dx, x.grad, BX3_TOL, f"bf16x3 dgrad {Cin}->{Cout}@{H}")


@pytest.mark.parametrize("B,Cin,Cout,H,fold", [(64, 128, 128, 32, True), (128, 256, 256, 16, True), (128, 256, 200, 16, False)])
def test_conv_epilogue_channel_sums_give_the_groupnorm_statistics(B, Cin, Cout, H, fold):
    """vd_gemm_desc.gn_part (16x16x32 kernel): per-tile (sum, sum of squares) of the final output per channel, and
    vd_groupnorm_stats_from_partials == vd_groupnorm_stats of the tensor (ResnetBlock2D conv1 -> norm2 of the no-grad forward)."""
    x = torch.randn(B, Cin, H, H, generator=g(0)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).to(DEV)
    b = torch.randn(Cout, generator=g(2)).to(DEV)
    temb = torch.randn(B, Cout, generator=g(3)).to(DEV)
    wd = w.view(Cout, -1)
    pk = ops.conv3_pack_weights(wd, Cout, Cin)
    out = torch.empty(B, Cout, H, H, device=DEV)
    tiles = H * H // 256
    part = torch.full((B, tiles, Cout, 2), float("nan"), device=DEV)
    ss_in = None
    if fold:                                                      # the GroupNorm-folding loader (MODE 3), as the sampler runs it
        G1 = torch.randn(Cin, generator=g(4)).to(DEV) * 0.3 + 1
        B1 = torch.randn(Cin, generator=g(5)).to(DEV) * 0.3
        ss_in = torch.empty(B, Cin, 2, device=DEV)
        m_, r_ = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
        ops.groupnorm_stats(x, G1, B1, ss_in, m_, r_, 32, 1e-6)
    ops.conv3x3(x, wd, b, out, rowadd=temb, rowadd_bstride=Cout, gn_ss=ss_in, a_packed=pk, gn_part=part)
    assert ops.GN_PART_WRITTEN
    o64 = out.double().view(B, Cout, tiles, 256)
    want = torch.stack([o64.sum(-1), (o64 * o64).sum(-1)], -1).permute(0, 2, 1, 3)          # [B, tiles, Cout, 2]
    e = float((part.double() - want).abs().max() / want.abs().max())
    assert e < 1e-6, e
    G2 = torch.randn(Cout, generator=g(6)).to(DEV) * 0.3 + 1
    B2 = torch.randn(Cout, generator=g(7)).to(DEV) * 0.3
    groups = 8 if Cout % 32 else 32
    ss_a, ss_b = torch.empty(B, Cout, 2, device=DEV), torch.empty(B, Cout, 2, device=DEV)
    ma, ra, mb, rb = (torch.empty(B * groups, device=DEV) for _ in range(4))
    ops.groupnorm_stats(out, G2, B2, ss_a, ma, ra, groups, 1e-6)
    ops.groupnorm_stats_from_partials(part, tiles, G2, B2, ss_b, mb, rb, H * H, groups, 1e-6)
    check(mb, ma, 1e-5, "mean from conv-epilogue partials")
    check(rb, ra, 1e-5, "rstd from conv-epilogue partials")
    check(ss_b, ss_a, 1e-5, "scale / shift from conv-epilogue partials")
    # a launch that does not go to the 16x16x32 kernel leaves the buffer alone and says so
    small = torch.empty(2, Cout, H, H, device=DEV)
    ops.conv3x3(x[:2], wd, b, small, a_packed=pk, gn_part=part)
    assert not ops.GN_PART_WRITTEN


@pytest.mark.parametrize("H,pad", [(32, 1), (16, 1), (8, 1), (8, 0), (128, 0), (128, 1), (256, 0), (256, 1)])
def test_split_precision_stride2_conv_with_symmetric_padding(H, pad):
    """Downsample2D(padding=1) (the LDM / VQ-VAE variant) and padding=0 (F.pad (0,1,0,1), the DDPM UNets) on the stride-2 patch;
    128 / 256-pixel inputs (round 4): the 64 / 128-pixel row-segment tiles of BASELINE config #4's first two Downsample2D layers."""
    B, Cin, Cout = (5, 64, 96) if H <= 32 else (2, 64, 96)
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g(2))
    y_ref = F.conv2d(x, w, b, stride=2, padding=1) if pad else F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    assert ops.bx3_eligible(Cout, Cin, H // 2, H // 2, B_CONV3_S2)
    wd = w.to(DEV).view(Cout, -1)
    out = torch.empty(B, Cout, H // 2, H // 2, device=DEV)
    ops.conv3x3(x.to(DEV), wd, b.to(DEV), out, mode=B_CONV3_S2, pad=pad, a_packed=ops.conv3_pack_weights(wd, Cout, Cin))
    check(out, y_ref, BX3_TOL, f"bf16x3 stride-2 conv pad={pad} @{H}")


@pytest.mark.parametrize("B,Cin,Cout,H,mode", [(64, 128, 128, 32, B_CONV3), (40, 192, 200, 32, B_CONV3), (128, 256, 256, 16, B_CONV3), (72, 64, 256, 16, B_CONV3),
                                               (64, 128, 128, 16, B_CONV3_UP), (128, 256, 128, 8, B_CONV3_UP)])
def test_split_precision_conv3x3_on_unsplit_grids(B, Cin, Cout, H, mode):
    """Full-size grids (>= 256 tiles of 128 channels x 128 pixels, no split-K): the shapes the training / sampling steps actually launch
    at 16x16 / 32x32 -- forward with the fused epilogue, the GroupNorm-folded loader, and the stride-1 input gradient."""
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    temb = torch.randn(B, Cout, generator=g(3))
    y0 = ref_conv(x, w, b, mode)
    res = torch.randn(y0.shape, generator=g(4))
    y_ref = y0 + temb[:, :, None, None] + res
    OH = y_ref.shape[-1]
    wd = w.detach().to(DEV).view(Cout, -1)
    pk = ops.conv3_pack_weights(wd, Cout, Cin)
    xd = x.detach().to(DEV)
    out = torch.empty(B, Cout, OH, OH, device=DEV)
    ops.conv3x3(xd, wd, b.to(DEV), out, mode=mode, rowadd=temb.to(DEV), rowadd_bstride=Cout, residual=res.to(DEV), a_packed=pk)
    check(out, y_ref.detach(), BX3_TOL, f"bf16x3 conv (unsplit grid) mode={mode} {Cin}->{Cout}@{H} B={B}")
    if mode == B_CONV3:
        # GroupNorm + SiLU folded into the loader == the two-kernel sequence, bit for bit
        gamma, beta = torch.rand(Cin, generator=g(5)) + 0.5, torch.randn(Cin, generator=g(6)) * 0.1
        G = 32 if Cin % 32 == 0 else 8
        a = torch.empty_like(xd)
        mean, rstd = torch.empty(B * G, device=DEV), torch.empty(B * G, device=DEV)
        ops.groupnorm_fwd(xd, gamma.to(DEV), beta.to(DEV), a, mean, rstd, G, 1e-6, True)
        ss = torch.empty(B, Cin, 2, device=DEV)
        ops.groupnorm_stats(xd, gamma.to(DEV), beta.to(DEV), ss, mean, rstd, G, 1e-6)
        o3, o4 = torch.empty_like(out), torch.empty_like(out)
        ops.conv3x3(xd, wd, b.to(DEV), o3, gn_ss=ss, a_packed=pk)
        ops.conv3x3(a, wd, b.to(DEV), o4, a_packed=pk)
        assert torch.equal(o3, o4), float((o3 - o4).abs().max())
        if Cout % 16:
            return                                              # the transposed operand needs 16-channel chunks of the OUTPUT side
        dy = torch.randn(y0.shape, generator=g(7))
        y0.backward(dy)
        pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True)
        wt = torch.empty(Cin, Cout * 9, device=DEV)
        dx = torch.empty(B, Cin, H, H, device=DEV)
        ops.conv3x3(dy.to(DEV), wt, None, dx, mode=B_CONV3_T, a_packed=pkt)
        check(dx, x.grad, BX3_TOL, f"bf16x3 dgrad (unsplit grid) {Cin}->{Cout}@{H}")
        dx2 = torch.empty_like(dx)                              # determinism: same launch, same bits
        ops.conv3x3(dy.to(DEV), wt, None, dx2, mode=B_CONV3_T, a_packed=pkt)
        assert torch.equal(dx, dx2)


@pytest.mark.parametrize("B,Cin,Cout,H", [(32, 256, 256, 16), (128, 128, 256, 8), (40, 200, 128, 16), (128, 64, 64, 8),
                                         (128, 256, 128, 8)])       # the first and the last land on the 128 x 256 tile (32x32 / 16x16 outputs)
def test_upsample_conv_input_gradient_with_the_2x2_sum_in_the_epilogue(B, Cin, Cout, H):
    """Upsample2D = nearest 2x + conv3x3; its input gradient is the stride-1 dgrad at the OUTPUT resolution followed by 2x2 block sums.
    vd_gemm_desc.pool2 does the sums in the dgrad epilogue (H is the input side; outputs 16x16 / 32x32, unsplit grid)."""
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9))
    y = ref_conv(x, w, None, B_CONV3_UP)
    dy = torch.randn(y.shape, generator=g(2))
    y.backward(dy)
    OH = 2 * H
    assert ops.bx3_pool2_eligible(Cin, Cout, OH, OH, B)
    wd = w.to(DEV).view(Cout, -1)
    pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True)
    wt = torch.empty(Cin, Cout * 9, device=DEV)
    dxbuf = torch.full((B, Cin + 2, H, H), 7.0, device=DEV)             # a channel slice of a wider buffer
    ops.conv3x3(dy.to(DEV), wt, None, dxbuf[:, 1:1 + Cin], mode=B_CONV3_T, a_packed=pkt, pool2=True)
    check(dxbuf[:, 1:1 + Cin], x.grad, BX3_TOL, f"bf16x3 upsample dgrad (fused 2x2 sum) {Cin}->{Cout}@{H}")
    assert float((dxbuf[:, 0] - 7).abs().max()) == 0 and float((dxbuf[:, -1] - 7).abs().max()) == 0
    dU, dx2 = torch.empty(B, Cin, OH, OH, device=DEV), torch.empty(B, Cin, H, H, device=DEV)
    ops.conv3x3(dy.to(DEV), wt, None, dU, mode=B_CONV3_T, a_packed=pkt)
    ops.sumpool2x2(dU, dx2)
    assert float((dxbuf[:, 1:1 + Cin] - dx2).abs().max()) <= 2e-6 * float(dx2.abs().max())      # same products, the four sums in another order
    # outside the supported set the flag must fail loudly, never fall back silently
    assert not ops.bx3_pool2_eligible(Cin, Cout, OH, OH, 1)
    with pytest.raises(RuntimeError):
        ops.conv3x3(dy[:1].to(DEV), wt, None, dx2[:1], mode=B_CONV3_T, a_packed=pkt, pool2=True)


@pytest.mark.parametrize("B,Cin,Cout,H,mode", [(4, 128, 128, 32, B_CONV3), (3, 192, 64, 32, B_CONV3), (2, 256, 256, 16, B_CONV3),
                                               (5, 64, 200, 16, B_CONV3), (6, 256, 128, 8, B_CONV3), (1, 64, 64, 8, B_CONV3),
                                               (128, 128, 128, 8, B_CONV3), (3, 128, 128, 16, B_CONV3_UP), (2, 256, 64, 8, B_CONV3_UP),
                                               (5, 64, 96, 4, B_CONV3_UP),
                                               # wide images: 32-pixel row segments, halo pixels from the neighbouring segments
                                               (2, 64, 128, 64, B_CONV3), (1, 72, 64, 128, B_CONV3), (1, 64, 64, 96, B_CONV3), (2, 64, 64, 32, B_CONV3_UP), (1, 80, 64, 64, B_CONV3_UP),
                                               # 4x4: two whole images per K-step (odd batch: the last step is half empty)
                                               (128, 256, 256, 4, B_CONV3), (7, 64, 96, 4, B_CONV3), (1, 128, 64, 4, B_CONV3)])
def test_split_precision_weight_gradient(B, Cin, Cout, H, mode):
    """H is the INPUT side; B_CONV3_UP: the weight gradient through the fused nearest-2x upsample (output 2H x 2H)."""
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    y = ref_conv(x, w, None, mode)
    dy = torch.randn(y.shape, generator=g(2))
    y.backward(dy)
    OH = y.shape[-1]
    assert ops.wgrad_bx3_eligible(Cout, Cin, OH, OH, mode)
    xbuf = torch.zeros(B, Cin + 4, H, H, device=DEV)
    xbuf[:, 4:] = x.to(DEV)
    need = ops.wgrad_ws_floats(Cout, Cin, 9, B, OH * OH, mode=mode, math_mode=1)
    ws = torch.empty(max(need, 4), device=DEV)
    dw = torch.full((Cout, Cin * 9), 0.5, device=DEV)
    ops.conv_wgrad(dy.to(DEV), xbuf[:, 4:], dw, mode, ws, accumulate=True, math_mode=1)
    check(dw - 0.5, w.grad.view(Cout, -1), BX3_TOL, f"bf16x3 wgrad mode={mode} {Cin}->{Cout}@{H} (ws {need})")
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(dy.to(DEV), xbuf[:, 4:], dw2, mode, ws, accumulate=False, splits=1, math_mode=1)
    check(dw2, w.grad.view(Cout, -1), BX3_TOL, "bf16x3 wgrad splits=1")
    dw3 = torch.empty_like(dw)                                  # deterministic: same launch, same bits
    ops.conv_wgrad(dy.to(DEV), xbuf[:, 4:], dw3, mode, ws, accumulate=False, math_mode=1)
    ops.conv_wgrad(dy.to(DEV), xbuf[:, 4:], dw2, mode, ws, accumulate=False, math_mode=1)
    assert torch.equal(dw2, dw3)


GROUPS = [
    # (input side H, mode, [(B, Cin, Cout), ...]) -- one kernel class per group, ragged shapes, mixed batch sizes
    (16, B_CONV3, [(16, 256, 256), (16, 512, 256), (5, 64, 200), (16, 128, 96)]),
    (32, B_CONV3, [(8, 128, 128), (8, 384, 128), (3, 192, 64)]),
    (8, B_CONV3, [(32, 256, 256), (6, 256, 128), (1, 64, 64)]),
    (4, B_CONV3, [(128, 256, 256), (7, 64, 96), (1, 128, 64)]),
    (8, B_CONV3_UP, [(8, 256, 128), (3, 128, 128)]),
    (64, B_CONV3, [(2, 64, 128), (1, 72, 64)]),
    (16, B_PLAIN, [(16, 256, 768), (16, 512, 256), (3, 80, 64), (16, 256, 256)]),
    (32, B_PLAIN, [(8, 384, 128), (8, 256, 128), (2, 128, 256)]),
    # the headline batch: at B = 128 a layer has K = 131 072 (32x32) / 32 768 (16x16) pixels and is split over 3-12 K ranges of <= 128 K-steps
    (32, B_CONV3, [(128, 128, 128), (128, 384, 128), (128, 256, 128)]),
    (16, B_CONV3, [(128, 256, 256), (128, 512, 256), (128, 128, 256)]),
    (32, B_PLAIN, [(128, 384, 128), (128, 256, 128)]),
    (16, B_PLAIN, [(128, 256, 768), (128, 512, 256)]),
    # tiny images: a 32-pixel K-step spans several images and the last step is partly empty (regression: its invalid octets must not address an
    # image in front of the step's descriptor base)
    (4, B_PLAIN, [(3, 128, 64), (5, 256, 192), (7, 64, 64)]),
    (8, B_PLAIN, [(3, 64, 128), (1, 320, 72)]),
]


@pytest.mark.parametrize("H,mode,jobs", GROUPS)
def test_grouped_weight_gradients(H, mode, jobs):
    """vd_conv_wgrad_group_*: several weight gradients of one kernel class in ONE launch pair, each job split over only as many
    workgroups as its share of the grid -- against torch's fp32 weight gradient and against the one-launch-per-convolution path."""
    T = 1 if mode == B_PLAIN else 9
    descs, keep, refs, outs = [], [], [], []
    for k, (B, Cin, Cout) in enumerate(jobs):
        x = torch.randn(B, Cin, H, H, generator=g(10 * k))
        w = (torch.randn(Cout, Cin, 3 if T == 9 else 1, 3 if T == 9 else 1, generator=g(10 * k + 1)) / math.sqrt(Cin * T)).requires_grad_()
        y = ref_conv(x, w, None, mode) if T == 9 else F.conv2d(x, w)
        dy = torch.randn(y.shape, generator=g(10 * k + 2))
        y.backward(dy)
        xbuf = torch.zeros(B, Cin + 4, H, H, device=DEV)            # operands that are channel slices of wider buffers
        xbuf[:, 4:] = x.to(DEV)
        dyd = dy.to(DEV)
        dw = torch.full((Cout, Cin * T), 0.25, device=DEV)
        d = ops.wgrad_desc(dyd, xbuf[:, 4:], dw, mode, None, accumulate=True, math_mode=1)
        assert ops.wgrad_group_class(d) != 0, (H, mode, B, Cin, Cout)
        descs.append(d)
        keep.append((xbuf, dyd))
        refs.append(w.grad.view(Cout, -1))
        outs.append(dw)
    assert len({ops.wgrad_group_class(d) for d in descs}) == 1
    ops.conv_wgrad_group(descs, torch.device(DEV))
    for k, (dw, ref) in enumerate(zip(outs, refs)):
        check(dw - 0.25, ref, BX3_TOL, f"grouped wgrad mode={mode} job {k} {jobs[k]}@{H}")
    snap = [o.clone() for o in outs]
    ops.conv_wgrad_group(descs, torch.device(DEV))                  # cached job table; accumulate = True adds the same bits again
    for o, s0, ref in zip(outs, snap, refs):
        check(o - s0, ref, BX3_TOL, "grouped wgrad, second accumulation")
    # the one-launch-per-convolution path computes the same sums in another split order
    for k, (B, Cin, Cout) in enumerate(jobs):
        xbuf, dyd = keep[k]
        OH = dyd.shape[-1]
        ws = torch.empty(max(ops.wgrad_ws_floats(Cout, Cin, T, B, OH * OH, mode=mode, math_mode=1), 4), device=DEV)
        single = torch.zeros(Cout, Cin * T, device=DEV)
        ops.conv_wgrad(dyd, xbuf[:, 4:], single, mode, ws, accumulate=False, math_mode=1)
        assert float((single - (snap[k] - 0.25)).abs().max()) <= 2e-5 * float(single.abs().max())
    # mixing kernel classes in one group is an error
    if mode == B_CONV3 and H == 16:
        x8, dy8, dw8 = torch.randn(2, 64, 8, 8, device=DEV), torch.randn(2, 64, 8, 8, device=DEV), torch.zeros(64, 64 * 9, device=DEV)
        with pytest.raises(RuntimeError):
            ops.conv_wgrad_group(descs[:1] + [ops.wgrad_desc(dy8, x8, dw8, B_CONV3, None, accumulate=True, math_mode=1)], torch.device(DEV))


def test_split_precision_requests_outside_the_supported_set_fail_loudly():
    from villandiffusion_amd.lib import VillanHipError
    x = torch.randn(2, 24, 16, 16, device=DEV)
    w = torch.randn(64, 24 * 9, device=DEV)
    pk = torch.zeros(128 * 32 * 9, device=DEV, dtype=torch.int32)
    with pytest.raises(VillanHipError):                                      # C % 16 != 0
        ops.conv3x3(x, w, None, torch.empty(2, 64, 16, 16, device=DEV), a_packed=pk)
    x2 = torch.randn(2, 64, 24, 24, device=DEV)
    with pytest.raises(VillanHipError):                                      # 24x24 image: not a split-precision tile size
        ops.conv_wgrad(torch.randn(2, 64, 24, 24, device=DEV), x2, torch.empty(64, 64 * 9, device=DEV), B_CONV3,
                       torch.empty(1 << 22, device=DEV), math_mode=1)


@pytest.mark.parametrize("B,Cin,Cout,H", [(16, 256, 768, 16), (8, 512, 256, 16), (4, 384, 128, 32), (12, 256, 200, 16), (3, 80, 64, 32),
                                          (5, 192, 96, 16), (6, 256, 128, 8), (16, 512, 256, 4)])   # 8x8 / 4x4: 2 / 8 whole images per tile
def test_split_precision_1x1_convolution_and_its_input_gradient(B, Cin, Cout, H):
    """gemm_bx3_kernel: W[M, C] @ x[b][C, HW] from the packed (hi, lo) weights; K = 80 / 200 exercise the tail stage (K % 64 != 0)."""
    x = torch.randn(B, Cin, H, H, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g(1)) / math.sqrt(Cin)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    res = torch.randn(B, Cout, H, H, generator=g(3))
    y0 = F.conv2d(x, w, b)
    y_ref = y0 + res
    assert ops.gemm_bx3_eligible(Cout, Cin, H * H, B)
    wd = w.detach().to(DEV).view(Cout, Cin)
    pk = ops.conv3_pack_weights(wd, Cout, Cin, taps=1)
    xbuf = torch.zeros(B, Cin + 2, H, H, device=DEV)
    xbuf[:, 2:] = x.detach().to(DEV)
    obuf = torch.full((B, Cout + 2, H, H), 7.0, device=DEV)
    ops.conv1x1(xbuf[:, 2:], wd, b.to(DEV), obuf[:, 1:1 + Cout], residual=res.to(DEV), a_packed=pk)
    check(obuf[:, 1:1 + Cout], y_ref.detach(), BX3_TOL, f"bf16x3 1x1 {Cin}->{Cout}@{H}")
    assert float((obuf[:, 0] - 7).abs().max()) == 0 and float((obuf[:, -1] - 7).abs().max()) == 0
    if Cout % 16 != 0 or Cin < 64:
        return
    dy = torch.randn(y0.shape, generator=g(4))
    y0.backward(dy)
    pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True, taps=1)
    dx = torch.empty(B, Cin, H, H, device=DEV)
    HW = H * H
    ops.gemm(wd, dy.to(DEV), dx, M=Cin, N=B * HW, K=Cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=Cin, ldb=HW, b_bstride=Cout * HW,
             ldd=HW, d_bstride=Cin * HW, a_packed=pkt)
    check(dx, x.grad, BX3_TOL, f"bf16x3 1x1 dgrad {Cin}->{Cout}@{H}")
    # weight gradient: both operands pixel-contiguous, K-steps of 64 pixels, deterministic slab reduction
    assert ops.wgrad_bx3_eligible(Cout, Cin, H, H, B_PLAIN)
    need = ops.wgrad_ws_floats(Cout, Cin, 1, B, HW, mode=B_PLAIN, math_mode=1)
    ws = torch.empty(max(need, 4), device=DEV)
    dw = torch.full((Cout, Cin), 0.25, device=DEV)
    ops.conv_wgrad(dy.to(DEV), xbuf[:, 2:], dw, B_PLAIN, ws, accumulate=True, math_mode=1)
    check(dw - 0.25, w.grad.view(Cout, Cin), BX3_TOL, f"bf16x3 1x1 wgrad {Cin}->{Cout}@{H} (ws {need})")
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(dy.to(DEV), xbuf[:, 2:], dw2, B_PLAIN, ws, accumulate=False, splits=1, math_mode=1)
    check(dw2, w.grad.view(Cout, Cin), BX3_TOL, "bf16x3 1x1 wgrad splits=1")


@pytest.mark.parametrize("a_row,b_kc", [(False, False), (True, False), (True, True), (False, True)])
def test_split_precision_activation_products(a_row, b_kc):
    """gemm_bx3_act_kernel: D[b] = alpha * A_b @ B_b with both operands split inside the kernel -- the four operand layouts of the
    attention contractions (scores: A m-contiguous, B n-contiguous; values: A k-contiguous; dv / dk: B k-contiguous), K = 80
    exercises the tail stage, per-batch strides larger than the matrices (q / k / v are channel slices of one tensor)."""
    nb, M, Nn, K = 5, 192, 256, (32 if a_row else 80) if not (a_row and b_kc) else 256        # K = 32: a multi-head score product (head_dim 32)
    gA = torch.randn(nb, 3, M * K, generator=g(0))          # operand = slice 1 of a wider buffer
    gB = torch.randn(nb, 2, K * Nn, generator=g(1))
    A = gA[:, 1].reshape(nb, M, K) if a_row else gA[:, 1].reshape(nb, K, M).transpose(1, 2)      # logical [nb, M, K]
    Bm = gB[:, 1].reshape(nb, Nn, K).transpose(1, 2) if b_kc else gB[:, 1].reshape(nb, K, Nn)    # logical [nb, K, Nn]
    ref = 0.37 * torch.bmm(A.double(), Bm.double()).float()
    dA, dB = gA.to(DEV), gB.to(DEV)
    D = torch.empty(nb, M, Nn, device=DEV)
    assert ops.gemm_bx3_act_eligible(M, K, Nn)
    ops.gemm(dA[:, 1], dB[:, 1], D, M=M, N=nb * Nn, K=K, a_mode=A_ROW if a_row else A_COL, b_mode=B_KCONTIG if b_kc else B_PLAIN, NP=Nn,
             lda=K if a_row else M, a_bstride=3 * M * K, ldb=K if b_kc else Nn, b_bstride=2 * K * Nn, ldd=Nn, d_bstride=M * Nn, alpha=0.37,
             math_mode=1)
    check(D, ref, BX3_TOL, f"bf16x3 activation product a_row={a_row} b_kcontig={b_kc}")
    D32 = torch.empty_like(D)
    ops.gemm(dA[:, 1], dB[:, 1], D32, M=M, N=nb * Nn, K=K, a_mode=A_ROW if a_row else A_COL, b_mode=B_KCONTIG if b_kc else B_PLAIN, NP=Nn,
             lda=K if a_row else M, a_bstride=3 * M * K, ldb=K if b_kc else Nn, b_bstride=2 * K * Nn, ldd=Nn, d_bstride=M * Nn, alpha=0.37)
    assert float((D - D32).abs().max()) <= 5e-5 * float(D32.std())


_TILE_PROBE = r"""
import hashlib, math, sys, torch
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T
g = torch.Generator().manual_seed(11)
out = []
for (cin, cout, H) in ((128, 128, 32), (256, 256, 16), (256, 256, 8)):
    x = torch.randn(128, cin, H, H, generator=g).cuda()
    w = (torch.randn(cout, cin * 9, generator=g) / math.sqrt(cin * 9)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    y = torch.empty(128, cout, H, H, device="cuda")
    ops.conv3x3(x, w, b, y, mode=B_CONV3, a_packed=ops.conv3_pack_weights(w, cout, cin))
    dx = torch.empty_like(x)
    ops.conv3x3(y, torch.empty(cin, cout * 9, device="cuda"), None, dx, mode=B_CONV3_T, a_packed=ops.conv3_pack_weights(w, cin, cout, transposed=True))
    torch.cuda.synchronize()
    out.append(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16] + hashlib.sha256(dx.cpu().numpy().tobytes()).hexdigest()[:16])
print("TILEHASH " + " ".join(out))
"""


def test_tile_choice_does_not_change_a_single_bit():
    """The 128x128, 128x256 and 128x512 tiles of the 32x32x16-MFMA kernel (and its split 128x256 tiles at 8x8) only regroup output elements over
    workgroups: every element is still accumulated chunk by chunk, tap by tap, in the same MFMA order, so the results are bit-identical whichever
    tile the planner picks (the planner is steered through its environment switches, which are read once per process: one subprocess per setting).
    The 16x16x32-MFMA kernels (vd_conv_k32.inc, and round 4's persistent vd_conv_k32p.hip, the default where it applies) contract 32 channels per instruction -- another summation grouping,
    so other bits, held to the same bounds against torch by the parity tests above; what is asserted for it here is run-to-run determinism."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = {}
    off = {"VD_BX3_K32_OFF": "1", "VD_K32P_OFF": "1"}
    for name, env in (("default", off), ("small", dict(off, VD_BX3_BIG_OFF="1", VD_BX3_BIGSPLIT_OFF="1")), ("no128x512", dict(off, VD_BX3_HUGE_OFF="1")),
                      ("k32", {"VD_K32P_OFF": "1"}), ("k32_again", {"VD_K32P_OFF": "1"}), ("k32p", {}), ("k32p_again", {})):
        e = dict(os.environ, PYTHONPATH=root, **env)
        r = subprocess.run([sys.executable, "-c", _TILE_PROBE], capture_output=True, text=True, env=e, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("TILEHASH")][0]
    print("[parity] output hashes per tile setting:", hashes)
    assert hashes["default"] == hashes["small"] == hashes["no128x512"]
    assert hashes["k32"] == hashes["k32_again"]
    # round 4: the persistent walk (vd_conv_k32p.hip: LDS-DMA weight stages, hand-pipelined fragment reads) keeps the 16x16x32 kernel's MFMA
    # order per output element (chunk pair by chunk pair, tap by tap): the same bits as round 3's kernel
    assert hashes["k32p"] == hashes["k32"] == hashes["k32p_again"]


_WK32_PROBE = r"""
import math, torch
import torch.nn.functional as F
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3, B_CONV3_UP
worst = 0.0
for (B, Cin, Cout, H, mode) in [(4, 128, 128, 32, B_CONV3), (3, 192, 64, 32, B_CONV3), (5, 64, 200, 16, B_CONV3), (6, 256, 128, 8, B_CONV3),
                                (128, 128, 128, 8, B_CONV3), (3, 128, 128, 16, B_CONV3_UP), (2, 256, 64, 8, B_CONV3_UP), (16, 384, 128, 32, B_CONV3)]:
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_()
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if mode == B_CONV3_UP else x
    y = F.conv2d(xin, w, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    OH = y.shape[-1]
    ws = torch.empty(max(ops.wgrad_ws_floats(Cout, Cin, 9, B, OH * OH, mode=mode, math_mode=1), 4), device="cuda")
    dw = torch.zeros(Cout, Cin * 9, device="cuda")
    ops.conv_wgrad(dy.cuda(), x.cuda(), dw, mode, ws, accumulate=False, math_mode=1)
    e = float((dw.cpu() - w.grad.view(Cout, -1)).abs().max() / w.grad.abs().max())
    worst = max(worst, e)
    if mode == B_CONV3 and H in (16, 32):          # the grouped launch: with VD_WGRAD9=1 the nine-taps-per-workgroup kernel
        dw2 = torch.zeros(Cout, Cin * 9, device="cuda")
        dyd, xd = dy.cuda(), x.cuda()
        d = ops.wgrad_desc(dyd, xd, dw2, mode, None, accumulate=True, math_mode=1)
        assert ops.wgrad_group_class(d) != 0
        ops.conv_wgrad_group([d], torch.device("cuda"))
        worst = max(worst, float((dw2.cpu() - w.grad.view(Cout, -1)).abs().max() / w.grad.abs().max()))
print("WK32 %.3e" % worst)
"""


def test_alternative_weight_gradient_kernels_stay_correct():
    """The 3x3 weight gradient has three kernels: vd_wgrad_k32.inc (16x16x32 MFMA, X staged once, two register sets: the default since round 4),
    wgrad_bx3_body (32x32x16, three shifted X copies: VD_WGRAD_K32=0) and vd_wgrad9.inc (all nine taps per workgroup: VD_WGRAD9=1).  The
    non-default ones stay built: hold each to the same bound in a process that selects it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sel in (dict(VD_WGRAD_K32="0"), dict(VD_WGRAD_K32="1", VD_WGRAD9="1")):
        e = dict(os.environ, PYTHONPATH=root, **sel)
        r = subprocess.run([sys.executable, "-c", _WK32_PROBE], capture_output=True, text=True, env=e, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        worst = float([ln for ln in r.stdout.splitlines() if ln.startswith("WK32")][0].split()[1])
        print(f"[parity] weight gradient with {sel}: worst rel_err {worst:.2e}")
        assert worst < BX3_TOL


_K32P_PROBE = r"""
import math, sys, torch
import torch.nn.functional as F
from villandiffusion_amd import ops
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T, B_CONV3_UP
worst = 0.0
def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
# (B, Cin, Cout, H, W, mode): 256-pixel tiles = 8 rows x 32 columns anywhere in the image (64 .. 256 wide, non-square, ragged channel tiles)
for (B, Cin, Cout, H, W, mode) in [(16, 64, 128, 64, 64, B_CONV3), (4, 64, 96, 128, 128, B_CONV3), (1, 32, 200, 256, 256, B_CONV3), (24, 64, 128, 40, 64, B_CONV3),
                                   (4, 64, 128, 64, 64, B_CONV3_UP), (2, 64, 64, 64, 128, B_CONV3_UP), (64, 128, 128, 32, 32, B_CONV3)]:
    g = torch.Generator().manual_seed(B * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == B_CONV3_UP else x
    y0 = F.conv2d(xin, w, b, padding=1)
    res = torch.randn(y0.shape, generator=g)
    y_ref = (y0 + temb[:, :, None, None] + res).detach()
    OH, OW = y0.shape[-2:]
    wd = w.cuda().view(Cout, -1)
    pk = ops.conv3_pack_weights(wd, Cout, Cin)
    xd = x.detach().cuda()
    out = torch.empty(B, Cout, OH, OW, device="cuda")
    part = torch.zeros(B, OH * OW // 256, Cout, 2, device="cuda")
    ops.conv3x3(xd, wd, b.cuda(), out, mode=mode, rowadd=temb.cuda(), rowadd_bstride=Cout, residual=res.cuda(), a_packed=pk, gn_part=part)
    assert ops.LAST_GEMM_TILE == 18, (ops.LAST_GEMM_TILE, B, Cin, Cout, H, W, mode)
    worst = max(worst, rel(out, y_ref))
    assert ops.GN_PART_WRITTEN                                  # per-tile channel sums of the FINAL result, fixed order
    s1 = part[..., 0].sum(1).cpu().double()
    worst = max(worst, float((s1 - y_ref.double().sum((2, 3))).abs().max() / y_ref.double().abs().sum((2, 3)).max()))
    s2 = part[..., 1].sum(1).cpu().double()
    worst = max(worst, float((s2 - (y_ref.double() ** 2).sum((2, 3))).abs().max() / (y_ref.double() ** 2).sum((2, 3)).max()))
    if mode == B_CONV3 and Cout % 32 == 0:
        dy = torch.randn(y0.shape, generator=g)
        y0.backward(dy)
        pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True)
        wt = torch.empty(Cin, Cout * 9, device="cuda")
        dx = torch.full((B, Cin, H, W), 3.0, device="cuda")
        ops.conv3x3(dy.cuda(), wt, None, dx, mode=B_CONV3_T, a_packed=pkt, accumulate=True)      # dx += ... (skip-gradient form)
        if ops.LAST_GEMM_TILE == 18:
            worst = max(worst, rel(dx - 3.0, x.grad))
    if mode == B_CONV3_UP and Cout % 32 == 0:                     # the upsample convolution's input gradient with the 2x2 sums in the epilogue
        dy = torch.randn(y0.shape, generator=g)
        y0.backward(dy)
        pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True)
        wt = torch.empty(Cin, Cout * 9, device="cuda")
        dxb = torch.full((B, Cin + 2, H, W), 7.0, device="cuda")
        ops.conv3x3(dy.cuda(), wt, None, dxb[:, 1:1 + Cin], mode=B_CONV3_T, a_packed=pkt, pool2=True)
        assert ops.LAST_GEMM_TILE == 18
        worst = max(worst, rel(dxb[:, 1:1 + Cin], x.grad))
        assert float((dxb[:, 0] - 7).abs().max()) == 0 and float((dxb[:, -1] - 7).abs().max()) == 0
print("K32P %.3e" % worst)
"""


def test_persistent_16x16x32_convolution_on_wide_and_non_square_images():
    """vd_conv_k32p.hip (round 4): the persistent tile walk of the 16x16x32 split-precision convolution on 64 .. 256-pixel-wide and non-square
    images (BASELINE configs #4 / #5: reference model.py:706-776), full epilogue (bias, time-embedding row, residual, accumulate, GroupNorm
    partial sums, the 2x2 sums of the upsample gradient) against torch's f32 convolution."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _K32P_PROBE], capture_output=True, text=True, env=e, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    worst = float([ln for ln in r.stdout.splitlines() if ln.startswith("K32P")][0].split()[1])
    print(f"[parity] persistent k32 convolution: worst rel_err {worst:.2e}")
    assert worst <= BX3_TOL


@pytest.mark.parametrize("B,Cin,Cout,H", [(128, 256, 768, 16), (128, 384, 128, 32), (128, 192, 200, 16), (512, 64, 256, 8), (8, 128, 128, 128)])
def test_persistent_16x16x32_1x1_convolution_and_its_input_gradient(B, Cin, Cout, H):
    """vd_gemm_k32p.hip (round 4): the persistent 16x16x32 kernel for the 1x1 family on full-size grids (vd_gemm_tile() == 19): the attention
    projection 256 -> 768, a shortcut at 32x32, a ragged channel tile (M = 200), tiles that span four 8x8 images, a 128x128 image (config #4);
    operands are channel slices of wider buffers; forward (+ bias + residual), accumulate, and the input gradient through the transposed
    packed operand, against torch."""
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = (torch.randn(Cout, Cin, 1, 1, generator=g(1)) / math.sqrt(Cin))
    b = torch.randn(Cout, generator=g(2))
    res = torch.randn(B, Cout, H, H, generator=g(3))
    wd = w.to(DEV).view(Cout, Cin)
    pk = ops.conv3_pack_weights(wd, Cout, Cin, taps=1)
    xbuf = torch.zeros(B, Cin + 2, H, H, device=DEV)
    xbuf[:, 2:] = x.to(DEV)
    obuf = torch.full((B, Cout + 2, H, H), 7.0, device=DEV)
    ops.conv1x1(xbuf[:, 2:], wd, b.to(DEV), obuf[:, 1:1 + Cout], residual=res.to(DEV), a_packed=pk)
    assert ops.LAST_GEMM_TILE == 19, ops.LAST_GEMM_TILE
    y_ref = torch.empty(B, Cout, H, H)
    for s0 in range(0, B, 64):                               # (the reference in chunks: the 128 x 128 case is 1 GB as one tensor on the host)
        y_ref[s0:s0 + 64] = F.conv2d(x[s0:s0 + 64], w, b) + res[s0:s0 + 64]
    check(obuf[:, 1:1 + Cout], y_ref, BX3_TOL, f"k32p 1x1 {Cin}->{Cout}@{H}")
    assert float((obuf[:, 0] - 7).abs().max()) == 0 and float((obuf[:, -1] - 7).abs().max()) == 0
    HW = H * H
    if Cout % 32 == 0 and Cin >= 64:
        dy = torch.randn(B, Cout, H, H, generator=g(4))
        pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True, taps=1)
        dx = torch.full((B, Cin, H, H), 2.0, device=DEV)
        ops.gemm(wd, dy.to(DEV), dx, M=Cin, N=B * HW, K=Cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=Cin, ldb=HW, b_bstride=Cout * HW,
                 ldd=HW, d_bstride=Cin * HW, a_packed=pkt, accumulate=True)
        if vd_cdiv_py(Cin, 128) * (B * HW // 256) >= 192:
            assert ops.LAST_GEMM_TILE == 19, ops.LAST_GEMM_TILE
        dx_ref = torch.empty(B, Cin, H, H)
        for s0 in range(0, B, 64):
            dx_ref[s0:s0 + 64] = F.conv_transpose2d(dy[s0:s0 + 64], w)
        check(dx - 2.0, dx_ref, BX3_TOL, f"k32p 1x1 dgrad {Cin}->{Cout}@{H}")
        dx2 = torch.full((B, Cin, H, H), 2.0, device=DEV)     # determinism
        ops.gemm(wd, dy.to(DEV), dx2, M=Cin, N=B * HW, K=Cout, a_mode=A_COL, b_mode=B_PLAIN, NP=HW, lda=Cin, ldb=HW, b_bstride=Cout * HW,
                 ldd=HW, d_bstride=Cin * HW, a_packed=pkt, accumulate=True)
        assert torch.equal(dx, dx2)


@pytest.mark.parametrize("B,Cin,Cout,H", [(8, 896, 2688, 8), (8, 2688, 896, 8), (8, 672, 672, 16), (2, 448, 1344, 32), (4, 320, 200, 8)])
def test_split_k_1x1_convolution_of_small_grids(B, Cin, Cout, H):
    """Round 4: 1x1 products with fewer than one round of 128 x 128 tiles and a long K (the attention projections of BASELINE config #5's 8x8 /
    16x16 levels at per-GPU batch 8) split K inside gemm_bx3_kernel<256> (vd_gemm_tile() == 9, vd_gemm_ws_floats() > 0) and add the slabs in fixed
    order: forward (+ bias + residual) and accumulate against torch, run-to-run identical; VD_GEMM_BX3_SPLIT_OFF=1 is the unsplit launch."""
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = (torch.randn(Cout, Cin, 1, 1, generator=g(1)) / math.sqrt(Cin))
    b = torch.randn(Cout, generator=g(2))
    res = torch.randn(B, Cout, H, H, generator=g(3))
    wd = w.to(DEV).view(Cout, Cin)
    pk = ops.conv3_pack_weights(wd, Cout, Cin, taps=1)
    out = torch.full((B, Cout, H, H), 7.0, device=DEV)
    ops.conv1x1(x.to(DEV), wd, b.to(DEV), out, residual=res.to(DEV), a_packed=pk)
    assert ops.LAST_GEMM_TILE == 9, ops.LAST_GEMM_TILE
    y_ref = F.conv2d(x, w, b) + res
    check(out, y_ref, BX3_TOL, f"split-K 1x1 {Cin}->{Cout}@{H}")
    out2 = torch.full((B, Cout, H, H), 7.0, device=DEV)
    ops.conv1x1(x.to(DEV), wd, b.to(DEV), out2, residual=res.to(DEV), a_packed=pk)
    assert torch.equal(out, out2)
    HW = H * H
    acc = torch.full((B, Cout, H, H), 2.0, device=DEV)
    ops.gemm(wd, x.to(DEV), acc, M=Cout, N=B * HW, K=Cin, a_mode=A_ROW, b_mode=B_PLAIN, NP=HW, lda=Cin, ldb=HW, b_bstride=Cin * HW, ldd=HW,
             d_bstride=Cout * HW, a_packed=pk, accumulate=True)
    check(acc - 2.0, F.conv2d(x, w), BX3_TOL, f"split-K 1x1 accumulate {Cin}->{Cout}@{H}")


def vd_cdiv_py(a, b):
    return (a + b - 1) // b


@pytest.mark.parametrize("B,Cin,Cout,H,pad", [(6, 128, 128, 32, 0), (128, 128, 128, 32, 0), (5, 256, 256, 16, 0), (3, 64, 200, 16, 1), (4, 192, 64, 32, 1),
                                              (64, 256, 256, 16, 0), (3, 64, 96, 64, 0), (2, 128, 64, 64, 1), (2, 64, 64, 128, 0), (2, 64, 128, 128, 1),
                                              (1, 64, 64, 256, 0), (2, 128, 128, 256, 1)])
def test_split_precision_weight_gradient_of_the_stride2_convolution(B, Cin, Cout, H, pad, monkeypatch):
    monkeypatch.setenv("VILLAN_WGRAD_S2", "all")             # (the default; ops._wgrad_s2_split)
    """Round 4: the Downsample2D convolution's weight gradient on the split-precision kernel (wgrad_bx3 MODE 4: the octet's 8 output columns read
    17 input columns, taps 0 / 2 share the even ones, tap 1 takes the odd ones) for both paddings (`F.pad (0,1,0,1)` of the DDPM UNets,
    `padding = 1` of the LDM / NCSN++ ones), single launch (accumulate, forced splits = 1) and the grouped launch, against torch; 64 .. 256-pixel
    inputs: 32x32 outputs and the 32-pixel row segments of wider ones (BASELINE config #4)."""
    x = torch.randn(B, Cin, H, H, generator=g(0))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    y = F.conv2d(x, w, None, stride=2, padding=1) if pad else F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2)
    dy = torch.randn(y.shape, generator=g(2))
    y.backward(dy)
    OH = H // 2
    assert ops.wgrad_bx3_eligible(Cout, Cin, OH, OH, B_CONV3_S2)
    xbuf = torch.zeros(B, Cin + 4, H, H, device=DEV)
    xbuf[:, 4:] = x.to(DEV)
    dyd = dy.to(DEV)
    need = ops.wgrad_ws_floats(Cout, Cin, 9, B, OH * OH, mode=B_CONV3_S2, math_mode=1)
    ws = torch.empty(max(need, 4), device=DEV)
    dw = torch.full((Cout, Cin * 9), 0.5, device=DEV)
    ops.conv_wgrad(dyd, xbuf[:, 4:], dw, B_CONV3_S2, ws, accumulate=True, pad=pad, math_mode=1)
    check(dw - 0.5, w.grad.view(Cout, -1), BX3_TOL, f"bf16x3 stride-2 wgrad pad={pad} {Cin}->{Cout}@{H} B={B} (ws {need})")
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(dyd, xbuf[:, 4:], dw2, B_CONV3_S2, ws, accumulate=False, splits=1, pad=pad, math_mode=1)
    check(dw2, w.grad.view(Cout, -1), BX3_TOL, "bf16x3 stride-2 wgrad splits=1")
    dw3 = torch.zeros_like(dw)
    d = ops.wgrad_desc(dyd, xbuf[:, 4:], dw3, B_CONV3_S2, None, accumulate=True, pad=pad, math_mode=1)
    assert ops.wgrad_group_class(d) == (2033 if OH >= 64 else 2000 + OH)
    ops.conv_wgrad_group([d], torch.device(DEV))
    check(dw3, w.grad.view(Cout, -1), BX3_TOL, "bf16x3 stride-2 wgrad (grouped launch)")
    # the exact-f32 kernel it replaces agrees as well (same reduction, other arithmetic)
    dw4 = torch.empty_like(dw)
    ws0 = torch.empty(max(ops.wgrad_ws_floats(Cout, Cin, 9, B, OH * OH, mode=B_CONV3_S2), 4), device=DEV)
    ops.conv_wgrad(dyd, xbuf[:, 4:], dw4, B_CONV3_S2, ws0, accumulate=False, pad=pad)
    assert float((dw4 - dw2).abs().max()) <= 2 * BX3_TOL * float(dw4.abs().max())
