"""PRE-SPLIT activation images (round 5; csrc/vd_presplit.hip, include/villan_hip.h): the producers write the bf16 (hi, lo) pairs the
split-precision kernels contract, the consumers fetch them without converting.

* the image format itself (pack / unpack against the definition written in torch);
* GroupNorm (+ SiLU) forward writing the image, against torch's fp32 group_norm + silu (the oracle's op sequence, oracle/unet_ref.py);
* the persistent 3x3 convolution (forward, flipped-tap input gradient, upsample-fused) reading it: BIT-IDENTICAL to the same kernel converting
  the same values itself;
* the grouped weight gradient with both operands pre-split (LDS-DMA + ds_read_b64_tr_b16): against torch's fp32 weight gradient and
  bit-identical to the converting 16x16x32 kernel (same products, same order).
Reference call sites replaced: F.group_norm / F.silu / F.conv2d inside diffusers ResnetBlock2D (reference loss.py:993) and the weight
gradients autograd derives for them (VillanDiffusion.py:1161)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.lib import B_CONV3, B_CONV3_T, B_CONV3_UP, VillanHipError  # noqa: E402

DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def split_ref(x):
    """The definition: hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def image_ref(x):
    """The pre-split image of x [B, C, H, W] as int16 [B, C/8, HW, 2, 8] (include/villan_hip.h)."""
    B, C, H, W = x.shape
    hi, lo = split_ref(x)
    def lay(t):
        return t.view(torch.int16).view(B, C // 8, 8, H * W).permute(0, 1, 3, 2)          # [B, o, p, 8]
    return torch.stack([lay(hi), lay(lo)], dim=3).contiguous()                              # [B, o, p, part, 8]


def test_pack_unpack_follow_the_format_definition():
    x = torch.randn(3, 24, 16, 16, generator=g(1)) * 3
    x[0, 0, 0, :4] = torch.tensor([0.0, 1e-30, -65504.0, 3.0e38])
    buf = torch.zeros(3, 32, 16, 16, device=DEV)                   # the image lives in a channel slice of a wider buffer (batch stride > C*HW)
    ps = ops.presplit_pack(x.to(DEV), out=ops.PreSplit(buf[:, 8:]))
    raw = buf.view(torch.int16).view(3, 32 // 8, 256, 2, 8)[:, 1:].cpu()
    assert torch.equal(raw, image_ref(x))
    assert float(buf[:, :8].abs().max()) == 0.0                    # nothing outside the slice was touched
    hi, lo = split_ref(x)
    back = ops.presplit_unpack(ps).cpu()
    assert torch.equal(back, hi.float() + lo.float())
    assert rel_err(back, x) < 2 ** -16
    with pytest.raises(AssertionError):
        ops.PreSplit(torch.zeros(1, 12, 4, 4, device=DEV))          # whole channel octets only


GN_SHAPES = [(4, 128, 32), (4, 256, 32), (3, 384, 32), (5, 256, 16), (4, 512, 16), (3, 384, 16), (6, 128, 16)]      # (B, C, side): 4 / 8 / 12 / 8 / 16 / 12 / 4 channels per group


@pytest.mark.parametrize("B,C,S", GN_SHAPES)
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_forward_writes_the_presplit_image(B, C, S, silu):
    G = 32
    assert ops.groupnorm_presplit_ok(C, S * S, G)
    x = torch.randn(B, C, S, S, generator=g(2)) * 1.7 + 0.4
    gamma = torch.randn(C, generator=g(3)) * 0.5 + 1
    beta = torch.randn(C, generator=g(4)) * 0.5
    ref = F.group_norm(x, G, gamma, beta, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    xd = torch.zeros(B, C + 8, S, S, device=DEV)                    # input and output both channel slices of wider buffers
    xd[:, 8:] = x.to(DEV)
    ybuf = torch.zeros(B, C + 16, S, S, device=DEV)
    y = ops.PreSplit(ybuf[:, 16:])
    mean, rstd = torch.empty(B * G, device=DEV), torch.empty(B * G, device=DEV)
    ops.groupnorm_fwd_presplit(xd[:, 8:], gamma.to(DEV), beta.to(DEV), y, mean, rstd, G, 1e-6, silu)
    got = ops.presplit_unpack(y)
    e = rel_err(got, ref)
    xg = x.view(B, G, -1)
    e_m = rel_err(mean.view(B, G), xg.mean(-1))
    e_r = rel_err(rstd.view(B, G), 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-6))
    print(f"[parity] GroupNorm{'+SiLU' if silu else ''} -> pre-split image B={B} C={C} {S}x{S}: output {e:.2e}, mean {e_m:.2e}, rstd {e_r:.2e}")
    assert e <= 3e-5 and e_m <= 1e-5 and e_r <= 1e-5              # the image keeps 16 significant bits: 2^-17 = 7.6e-6 per element
    assert float(ybuf[:, :16].abs().max()) == 0.0
    # the same values as the f32 kernel followed by the split (to the statistics' rounding: other summation order)
    yf = torch.empty(B, C, S, S, device=DEV)
    ops.groupnorm_fwd(xd[:, 8:], gamma.to(DEV), beta.to(DEV), yf, torch.empty_like(mean), torch.empty_like(rstd), G, 1e-6, silu)
    assert rel_err(got, yf) <= 2e-5
    assert not ops.groupnorm_presplit_ok(C, 64, G) and not ops.groupnorm_presplit_ok(C + 4, S * S, G)


@pytest.mark.parametrize("B,C,S", GN_SHAPES)
@pytest.mark.parametrize("silu,dual", [(True, False), (True, True), (False, True)])
def test_groupnorm_backward_writes_the_presplit_image(B, C, S, silu, dual):
    """gn_bwd_ps_kernel against autograd of torch's fp32 group_norm (+ silu): dx (pre-split image, and the f32 copy when both are asked for), the
    dgamma / dbeta rows summed over the batch, the residual terms extra / extra2 and the per-channel sums of dx -- and against the f32 kernel."""
    G = 32
    x = (torch.randn(B, C, S, S, generator=g(2)) * 1.7 + 0.4).requires_grad_()
    gamma = (torch.randn(C, generator=g(3)) * 0.5 + 1).requires_grad_()
    beta = (torch.randn(C, generator=g(4)) * 0.5).requires_grad_()
    y = F.group_norm(x, G, gamma, beta, eps=1e-6)
    if silu:
        y = F.silu(y)
    dy = torch.randn(B, C, S, S, generator=g(5))
    ex, ex2 = torch.randn(B, C, S, S, generator=g(6)), torch.randn(B, C, S, S, generator=g(7))
    y.backward(dy)
    dx_ref = x.grad + ex + ex2
    xd, dyd = x.detach().to(DEV), dy.to(DEV)
    mean, rstd = torch.empty(B * G, device=DEV), torch.empty(B * G, device=DEV)
    ops.groupnorm_fwd(xd, gamma.detach().to(DEV), beta.detach().to(DEV), torch.empty_like(xd), mean, rstd, G, 1e-6, silu)
    dxp = ops.presplit_empty(xd.shape, DEV)
    dxf = torch.empty_like(xd) if dual else None
    wg, wb = torch.empty(B * C, device=DEV), torch.empty(B * C, device=DEV)
    rs = torch.full((B, C + 3), float("nan"), device=DEV)
    ops.groupnorm_bwd_presplit(dyd, xd, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), dxf, dxp, wg, wb, G, silu, extra=ex.to(DEV),
                               extra2=ex2.to(DEV), rowsum=rs, rowsum_ld=C + 3)
    got = ops.presplit_unpack(dxp)
    e_dx = rel_err(got, dx_ref)
    e_g = rel_err(wg.view(B, C).sum(0), gamma.grad)
    e_b = rel_err(wb.view(B, C).sum(0), beta.grad)
    e_rs = rel_err(rs[:, :C], dx_ref.sum((2, 3)))
    print(f"[parity] GroupNorm{'+SiLU' if silu else ''} backward -> pre-split dx B={B} C={C} {S}x{S}: dx {e_dx:.2e}, dgamma {e_g:.2e}, dbeta {e_b:.2e}, row sums {e_rs:.2e}")
    assert e_dx <= 3e-5 and e_g <= 2e-5 and e_b <= 2e-5 and e_rs <= 2e-5
    assert bool(torch.isnan(rs[:, C:]).all())
    if dual:
        assert rel_err(dxf, dx_ref) <= 2e-5
        hi, lo = split_ref(dxf.cpu())
        assert torch.equal(got.cpu(), hi.float() + lo.float())           # the image IS the split of the f32 copy
    dx2 = torch.empty_like(xd)
    wg2, wb2 = torch.empty_like(wg), torch.empty_like(wb)
    ops.groupnorm_bwd(dyd, xd, mean, rstd, gamma.detach().to(DEV), beta.detach().to(DEV), dx2, wg2, wb2, G, silu, extra=ex.to(DEV), extra2=ex2.to(DEV))
    assert rel_err(got, dx2) <= 2e-5 and rel_err(wg, wg2) <= 1e-5 and rel_err(wb, wb2) <= 1e-5


CONV_PS = [
    # B, Cin, Cout, side of the OUTPUT, mode  -- batches that give the persistent kernel >= 192 tiles of 256 pixels (vd_gemm_tile() == 18)
    (48, 128, 128, 32, B_CONV3), (48, 384, 128, 32, B_CONV3), (48, 128, 256, 32, B_CONV3_T), (96, 256, 256, 16, B_CONV3), (96, 512, 256, 16, B_CONV3_T),
    (48, 256, 128, 32, B_CONV3_UP), (96, 256, 256, 16, B_CONV3_UP),
]


@pytest.mark.parametrize("B,Cin,Cout,S,mode", CONV_PS)
def test_persistent_convolution_reads_the_presplit_image_bit_for_bit(B, Cin, Cout, S, mode):
    """conv3_k32p_kernel<..., PS = true> copies the producer's (hi, lo) units where the PS = false kernel converts: same values, same LDS image,
    same MFMA order -> identical bits, for the forward, the flipped-tap input gradient and the upsample-fused forward, with bias + residual."""
    Sin = S // 2 if mode == B_CONV3_UP else S
    x = torch.randn(B, Cin, Sin, Sin, generator=g(5)).to(DEV)
    w = (torch.randn(Cout, Cin * 9, generator=g(6)) / math.sqrt(Cin * 9)).to(DEV)
    bias = torch.randn(Cout, generator=g(7)).to(DEV)
    res = torch.randn(B, Cout, S, S, generator=g(8)).to(DEV)
    pk = ops.conv3_pack_weights(w, Cout, Cin)      # (for the flipped-tap kind any packed [Cout, Cin * 9] operand serves: both runs read the same one)
    out_f = torch.empty(B, Cout, S, S, device=DEV)
    ops.conv3x3(x, w, bias, out_f, mode=mode, residual=res, a_packed=pk)
    assert ops.LAST_GEMM_TILE == 18
    out_p = torch.empty_like(out_f)
    ops.conv3x3(ops.presplit_pack(x), w, bias, out_p, mode=mode, residual=res, a_packed=pk)
    assert ops.LAST_GEMM_TILE == 18
    torch.cuda.synchronize()
    assert torch.equal(out_f, out_p), float((out_f - out_p).abs().max())
    assert float(out_p.abs().max()) > 0.1


def test_presplit_operand_outside_the_persistent_kernel_fails_loudly():
    x = torch.randn(2, 128, 32, 32, device=DEV)
    w = torch.randn(128, 128 * 9, device=DEV)
    pk = ops.conv3_pack_weights(w, 128, 128)
    with pytest.raises(VillanHipError):                             # 8 tiles: another kernel takes this problem, and it cannot read the image
        ops.conv3x3(ops.presplit_pack(x), w, None, torch.empty(2, 128, 32, 32, device=DEV), a_packed=pk)
    dy = torch.randn(2, 128, 32, 32, device=DEV)
    dw = torch.zeros(128, 128 * 9, device=DEV)
    with pytest.raises(VillanHipError):                             # the single-layer weight gradient has no pre-split kernel
        ops.conv_wgrad(ops.presplit_pack(dy), ops.presplit_pack(x), dw, B_CONV3, torch.empty(1 << 21, device=DEV), math_mode=1)
    d = ops.wgrad_desc(ops.presplit_pack(dy), x, dw, B_CONV3, None, accumulate=True, math_mode=1)
    assert ops.wgrad_group_class(d) == 0                            # one operand pre-split, the other not: no class


WGRAD_PS = [
    # side of the OUTPUT, mode, [(B, Cin, Cout), ...]: one grouped launch per row
    (32, B_CONV3, [(8, 128, 128), (8, 384, 128), (3, 192, 64)]),
    (16, B_CONV3, [(16, 256, 256), (16, 512, 256), (5, 64, 200), (16, 128, 96)]),
    (8, B_CONV3, [(32, 256, 256), (6, 256, 128), (1, 64, 64)]),
    (32, B_CONV3, [(128, 128, 128), (128, 384, 128), (128, 256, 128)]),       # the headline batch: K = 131 072 pixels, 3-12 K ranges per layer
    (16, B_CONV3, [(128, 256, 256), (128, 512, 256), (128, 128, 256)]),
    # the convolution behind Upsample2D: X is the half-resolution source, the doubling and the tap shift live in the transposed reads' row addresses
    (32, B_CONV3_UP, [(16, 256, 256), (3, 64, 72)]),
    (16, B_CONV3_UP, [(32, 256, 256), (5, 128, 64)]),
    (8, B_CONV3_UP, [(64, 256, 256), (7, 64, 128)]),
]


@pytest.mark.parametrize("S,mode,jobs", WGRAD_PS)
def test_grouped_weight_gradient_on_presplit_operands(S, mode, jobs):
    """wgrad_ps_group_kernel: both operands fetched by LDS-DMA, fragments through ds_read_b64_tr_b16 -- against torch's fp32 weight gradient and
    BIT-IDENTICAL to the converting kernel (wgrad_k32_group_kernel) on the same grouped plan."""
    d_ps, d_f, keep, refs, out_ps, out_f = [], [], [], [], [], []
    up = 2 if mode == B_CONV3_UP else 0
    Sx = S // 2 if up else S
    for k, (B, Cin, Cout) in enumerate(jobs):
        x = torch.randn(B, Cin, Sx, Sx, generator=g(20 * k))
        w = (torch.randn(Cout, Cin, 3, 3, generator=g(20 * k + 1)) / math.sqrt(Cin * 9)).requires_grad_()
        y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, w, None, padding=1)
        dy = torch.randn(y.shape, generator=g(20 * k + 2))
        y.backward(dy)
        xbuf = torch.zeros(B, Cin + 8, Sx, Sx, device=DEV)           # operands that are channel slices of wider buffers (skip concatenation)
        xbuf[:, 8:] = x.to(DEV)
        dyd = dy.to(DEV)
        xp_buf = torch.zeros(B, Cin + 8, Sx, Sx, device=DEV)
        xp = ops.presplit_pack(xbuf[:, 8:], out=ops.PreSplit(xp_buf[:, 8:]))
        dyp = ops.presplit_pack(dyd)
        dw_p = torch.full((Cout, Cin * 9), 0.25, device=DEV)
        dw_f = torch.full((Cout, Cin * 9), 0.25, device=DEV)
        dp = ops.wgrad_desc(dyp, xp, dw_p, mode, None, accumulate=True, math_mode=1)
        df = ops.wgrad_desc(dyd, xbuf[:, 8:], dw_f, mode, None, accumulate=True, math_mode=1)
        assert ops.wgrad_group_class(dp) == 3000 + 4 * S + up and ops.wgrad_group_class(df) == 4 * S + up, (ops.wgrad_group_class(dp), ops.wgrad_group_class(df))
        d_ps.append(dp)
        d_f.append(df)
        keep.append((xbuf, dyd, xp_buf, xp, dyp))
        refs.append(w.grad.view(Cout, -1))
        out_ps.append(dw_p)
        out_f.append(dw_f)
    ops.conv_wgrad_group(d_ps, torch.device(DEV))
    ops.conv_wgrad_group(d_f, torch.device(DEV))
    torch.cuda.synchronize()
    for k, (dp, df, ref) in enumerate(zip(out_ps, out_f, refs)):
        e = rel_err(dp - 0.25, ref)
        print(f"[parity] pre-split grouped wgrad (mode {mode}) job {k} {jobs[k]}@{S}: rel_err={e:.3e} vs torch; max |ps - converting| = {float((dp - df).abs().max()):.3e}")
        assert e <= 1e-4, (k, e)
        assert torch.equal(dp, df), (k, float((dp - df).abs().max()))
    snap = [o.clone() for o in out_ps]
    ops.conv_wgrad_group(d_ps, torch.device(DEV))                    # cached job table; accumulate = True adds the same bits again
    torch.cuda.synchronize()
    for o, s0, ref in zip(out_ps, snap, refs):
        assert rel_err(o - s0, ref) <= 1e-4


@pytest.mark.parametrize("B", [48, 64, 96])
def test_mixed_presplit_and_converting_blocks_hand_gradients_over_correctly(B):
    """Batches at which only SOME blocks take the pre-split path (the 16x16 level's grids are too small for the persistent kernel below B = 96):
    a converting block must still write the pre-split copy of its input gradient when the layer behind it asked for one (regression: the
    shortcut branch of the converting backward dropped it, and the Upsample2D weight gradient read an uninitialised image -> NaN at B = 64)."""
    from villandiffusion_amd.unet import UNet2DModel
    net = UNet2DModel()
    net.reset_parameters(seed=3)
    x = torch.randn(B, 3, 32, 32, generator=g(1)).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=g(2)).to(DEV)
    dy = (torch.randn(B, 3, 32, 32, generator=g(3)) * 1e-4).to(DEV)
    grads = {}
    for ps in (False, True):
        net.presplit = ps
        net.zero_grad()
        y = net(x, t, return_dict=False)[0]
        y.backward(dy)
        torch.cuda.synchronize()
        grads[ps] = net.flat_grad.detach().clone()
    assert bool(torch.isfinite(grads[True]).all())
    e = float((grads[True] - grads[False]).norm() / grads[False].norm())
    print(f"[parity] B={B}: pre-split vs converting gradients (L2) {e:.2e}")
    assert e <= 1e-4


@pytest.mark.timeout(900)
def test_network_step_with_presplit_operands_matches_the_converting_path():
    """BASELINE config #2's UNet at B = 128 (the batch whose grids the persistent kernels take): one poisoned-batch forward + backward with the
    producers writing pre-split images (default) against the same step with VILLAN_PRESPLIT=0 semantics (net.presplit = False: round 4's converting
    kernels).  The two differ only in the summation order of the GroupNorm statistics: loss and every parameter gradient agree to ~1e-6; and the
    pre-split kernels really ran (profile records), including gradients handed from block to block in both forms."""
    from villandiffusion_amd.loss import LossFn
    from villandiffusion_amd.schedulers import DDPMScheduler
    from villandiffusion_amd.unet import UNet2DModel
    B = 128
    gg = g(77)
    x0 = (torch.rand(B, 3, 32, 32, generator=gg) * 2 - 1).to(DEV)
    R = (torch.rand(B, 3, 32, 32, generator=gg) * 2 - 1).to(DEV)
    R[: B - B // 10] = 0
    eps = torch.randn(B, 3, 32, 32, generator=gg).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=gg).to(DEV)
    net = UNet2DModel()
    net.reset_parameters(seed=5)
    res = {}
    for ps in (True, False):
        net.presplit = ps
        lf = LossFn(DDPMScheduler(), "SDE-VP", psi=1)
        net.zero_grad()
        ops.profile_start()
        loss = lf.p_loss_by_keys({"target": x0, "pixel_values": R}, net, "target", "pixel_values", t, noise=eps)
        loss.backward()
        torch.cuda.synchronize()
        names = {r_["name"] for r_ in ops.profile_stop()}
        res[ps] = (float(loss), net.flat_grad.detach().clone(), names)
    (l1, g1, n1), (l0, g0, n0) = res[True], res[False]
    used = [n for n in n1 if "_ps_" in n or "presplit" in n or n.endswith("true>")]
    print(f"[parity] pre-split vs converting step at B=128: loss {abs(l1 - l0) / abs(l0):.2e}, gradient (L2) {float((g1 - g0).norm() / g0.norm()):.2e}, "
          f"worst element {float((g1 - g0).abs().max() / g0.abs().max()):.2e}; pre-split kernels: {sorted(used)}")
    assert any("wgrad_ps_group_kernel<32, 0>" in n for n in n1) and any("wgrad_ps_group_kernel<16, 0>" in n for n in n1)
    assert any("wgrad_ps_group_kernel<32, 2>" in n for n in n1)                 # the Upsample2D convolution (operands packed on the side stream)
    assert any("gn_fwd_ps_kernel" in n for n in n1) and any("gn_bwd_ps_kernel" in n for n in n1)
    assert any(n.startswith("conv3_k32p_kernel<32, 0") and n.endswith("true>") for n in n1) and any(n.startswith("conv3_k32p_kernel<16, 1") and n.endswith("true>") for n in n1)
    assert not any("_ps_" in n for n in n0)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert float((g1 - g0).norm() / g0.norm()) <= 2e-5 and float((g1 - g0).abs().max() / g0.abs().max()) <= 1e-4
