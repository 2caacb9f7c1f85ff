"""conv3_sm_kernel (vd_conv_sm.hip, round 6): the 8x8 / 4x4 levels' 3x3 convolutions with the WHOLE channel loop per workgroup (64-channel x 2 | 4-image
tiles, no split-K slabs, no epilogue launch) -- forward with the fused epilogue and the flipped-tap input gradient against torch's fp32 convolution
(the oracle's arithmetic) at the shapes the DDPM UNets launch and at ragged ones; tolerance as for every split-precision kernel (1e-4 of the output's
scale, ten times inside north_star's 1e-3) and <= 5e-5 of the standard deviation against the exact-f32 kernel."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T  # noqa: E402

DEV = "cuda"
TOL = 1e-4


def g(seed):
    return torch.Generator().manual_seed(seed)


def check(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    print(f"[parity] {what}: rel_err={e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{what}: {e:.3e} > {tol:.1e}"


CASES = [
    # B, Cin, Cout, S: the training / sampling shapes of config #2 (B = 128), then ragged batches (a last tile with fewer images), ragged M, small / large C.
    # The kernel takes 8x8 grids of >= 64 tiles and 4x4 grids of >= 256 tiles (vd_conv3_sm_eligible): every case is one in the forward direction.
    (128, 256, 256, 8), (128, 512, 256, 8), (128, 256, 512, 4), (256, 512, 256, 4),
    (33, 256, 256, 8), (35, 64, 200, 8), (205, 96, 320, 4), (1024, 32, 64, 4), (16, 1024, 512, 8),
]


def takes(M, B, S):
    return M >= 64 and -(-M // 64) * -(-B // (2 if S == 8 else 4)) >= (64 if S == 8 else 256)


@pytest.mark.parametrize("B,Cin,Cout,S", CASES)
def test_whole_k_convolution_forward_epilogue_and_input_gradient(B, Cin, Cout, S):
    x = torch.randn(B, Cin, S, S, generator=g(0), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(1)) / math.sqrt(Cin * 9)).requires_grad_()
    b = torch.randn(Cout, generator=g(2))
    temb = torch.randn(B, Cout + 5, generator=g(3))
    y0 = F.conv2d(x, w, b, padding=1)
    res = torch.randn(y0.shape, generator=g(4))
    y_ref = (y0 + temb[:, 2:2 + Cout, None, None] + res).detach()
    wd = w.detach().to(DEV).view(Cout, -1)
    pk = ops.conv3_pack_weights(wd, Cout, Cin)
    xbuf = torch.zeros(B, Cin + 3, S, S, device=DEV)            # channel slices of wider buffers (the zero-copy skip concatenation)
    xbuf[:, 3:] = x.detach().to(DEV)
    obuf = torch.full((B, Cout + 8, S, S), 7.0, device=DEV)
    ops.conv3x3(xbuf[:, 3:], wd, b.to(DEV), obuf[:, 4:4 + Cout], rowadd=temb.to(DEV)[:, 2:], rowadd_bstride=Cout + 5, residual=res.to(DEV), a_packed=pk)
    assert takes(Cout, B, S) and ops.LAST_GEMM_TILE == 20, ops.LAST_GEMM_TILE
    check(obuf[:, 4:4 + Cout], y_ref, TOL, f"whole-K conv {Cin}->{Cout}@{S} B={B}")
    assert float((obuf[:, :4] - 7).abs().max()) == 0 and float((obuf[:, 4 + Cout:] - 7).abs().max()) == 0      # nothing outside the slice
    o32 = torch.empty(B, Cout, S, S, device=DEV)
    ops.conv3x3(xbuf[:, 3:], wd, b.to(DEV), o32, rowadd=temb.to(DEV)[:, 2:], rowadd_bstride=Cout + 5, residual=res.to(DEV))       # exact-f32 kernel
    assert float((obuf[:, 4:4 + Cout] - o32).abs().max()) <= 5e-5 * float(o32.std())
    # accumulate: D += W (*) x
    acc = res.to(DEV).clone()
    ops.conv3x3(xbuf[:, 3:], wd, None, acc, accumulate=True, a_packed=pk)
    assert ops.LAST_GEMM_TILE == 20
    check(acc, (F.conv2d(x, w, None, padding=1) + res).detach(), TOL, "whole-K conv, accumulate")
    if Cout % 32 != 0 or Cin < 64:                               # (no split-precision kernel takes fewer than 64 output rows: the input gradient of Cin < 64)
        return
    # input gradient: flipped taps over the transposed packed operand (K = Cout)
    dy = torch.randn(y0.shape, generator=g(5))
    y0.backward(dy)
    pkt = ops.conv3_pack_weights(wd, Cin, Cout, transposed=True)
    wt = torch.empty(Cin, Cout * 9, device=DEV)                  # shape carrier only: the kernel reads pkt
    dx = torch.empty(B, Cin, S, S, device=DEV)
    ops.conv3x3(dy.to(DEV), wt, None, dx, mode=B_CONV3_T, a_packed=pkt)
    assert (ops.LAST_GEMM_TILE == 20) == takes(Cin, B, S), ops.LAST_GEMM_TILE      # (smaller grids stay on the split kernels)
    check(dx, x.grad, TOL, f"whole-K input gradient {Cout}->{Cin}@{S}")


def test_small_grids_keep_the_split_kernels():
    """Fewer than 64 (8x8) / 256 (4x4) tiles -- small batches: most of the other GPU tests -- stay on the split-K kernels of rounds 2-5; so does everything
    the whole-K kernel does not read (16x16 and larger images)."""
    x = torch.randn(4, 256, 8, 8, device=DEV)
    w = torch.randn(256, 256 * 9, device=DEV) / 48
    out = torch.empty(4, 256, 8, 8, device=DEV)
    pk = ops.conv3_pack_weights(w, 256, 256)
    ops.conv3x3(x, w, None, out, a_packed=pk)
    assert ops.LAST_GEMM_TILE != 20
    x4 = torch.randn(128, 256, 4, 4, device=DEV)
    ops.conv3x3(x4, w, None, torch.empty(128, 256, 4, 4, device=DEV), a_packed=pk)
    assert ops.LAST_GEMM_TILE != 20                              # 4x4 at M = 256, B = 128: 128 tiles
    x16 = torch.randn(128, 256, 16, 16, device=DEV)
    o16 = torch.empty(128, 256, 16, 16, device=DEV)
    ops.conv3x3(x16, w, None, o16, a_packed=pk)
    assert ops.LAST_GEMM_TILE == 18
