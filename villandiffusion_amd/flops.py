"""Analytic (algorithmic, 2 x MAC) forward FLOPs per image of a native UNet2DModel, walked from its layer objects -- the number the
roofline fractions of bench.py are quoted on (12.444 GFLOP for the DDPM-CIFAR10-32 UNet; training = 3 x forward).
Counted: every 3x3 / 1x1 convolution and linear layer, the attention products QK^T and PV.  Not counted (as in SURVEY's figure):
GroupNorm, SiLU, softmax, residual adds, bias adds, the sinusoidal embedding."""
from __future__ import annotations

from typing import Dict


def unet_forward_flops(net, breakdown: bool = False):
    S = net.sample_size
    f: Dict[str, float] = {"conv3x3": 0.0, "upsample_conv": 0.0, "downsample_conv": 0.0, "conv1x1_shortcut": 0.0, "attn_proj": 0.0,
                           "attn_core": 0.0, "linear": 0.0, "conv_in_out": 0.0}

    def res(r, hw):
        f["conv3x3"] += 2.0 * 9 * (r.cin * r.cout + r.cout * r.cout) * hw
        if r.has_sc:
            f["conv1x1_shortcut"] += 2.0 * r.cin * r.cout * hw
        f["linear"] += 2.0 * net.temb_dim * r.cout                      # time_emb_proj row of this block

    def attn(a, hw):
        f["attn_proj"] += 2.0 * (3 * a.ch * a.ch + a.ch * a.ch) * hw    # to_q / to_k / to_v + to_out
        f["attn_core"] += 2.0 * 2 * hw * hw * a.ch                      # QK^T and PV over all heads (heads split the channels)

    f["linear"] += 2.0 * (net.time_dim0 * net.temb_dim + net.temb_dim * net.temb_dim)
    f["conv_in_out"] += 2.0 * 9 * net.in_channels * net._conv_in.cout * S * S
    s = S
    for blk in net.down:
        for j, r in enumerate(blk["res"]):
            res(r, s * s)
            if blk["attn"]:
                attn(blk["attn"][j], s * s)
        if blk["ds"] is not None:
            s //= 2
            f["downsample_conv"] += 2.0 * 9 * blk["ds"].cin * blk["ds"].cout * s * s
    for r in net.mid_res:
        res(r, s * s)
    attn(net.mid_attn, s * s)
    for blk in net.up:
        for j, r in enumerate(blk["res"]):
            res(r, s * s)
            if blk["attn"]:
                attn(blk["attn"][j], s * s)
        if blk["us"] is not None:
            s *= 2
            f["upsample_conv"] += 2.0 * 9 * blk["us"].cin * blk["us"].cout * s * s
    f["conv_in_out"] += 2.0 * 9 * net._conv_out.cin * net.out_channels * S * S
    total = sum(f.values())
    return (total, f) if breakdown else total


# ---- which matrix pipe a kernel symbol runs on (bench.py prices a kernel against the dense peak of the arithmetic it executes) ----
# Split-precision kernels: every algorithmic product term is three bf16 MFMAs (hi*hi + hi*lo + lo*hi) -> 2500 TFLOP/s dense bf16 peak on
# ALGORITHMIC FLOPs, x3 for the executed fraction.  Everything else that `ops` records with kind "mfma" runs on v_mfma_f32_32x32x2_f32 (157.3).
SPLIT_PRECISION_FAMILIES = ("bx3", "attn_core", "attn_flash", "k32", "wgrad9", "wgrad1x1_wide", "wgrad_ps", "presplit", "conv3_sm")
EXACT_F32_FAMILIES = ("gemm_kernel<", "gemm_plain_kernel", "conv3_patch_kernel", "wgrad_patch_kernel", "wgrad_patch_gen_kernel", "wgrad_kernel<",
                      "wgrad_small_kernel", "conv3_fewout_kernel", "conv3_smallm_kernel", "attn_small")


def is_split_precision(kernel_name: str) -> bool:
    """True for kernels of the split-precision (bf16 x 3) arithmetic; raises for a matrix kernel that is in neither list, so a new kernel cannot be
    priced against the wrong peak silently (round 4 priced `attn_flash_kernel` against the f32 peak: 0.79 printed, 0.05 true)."""
    if any(f in kernel_name for f in SPLIT_PRECISION_FAMILIES):
        return True
    if any(f in kernel_name for f in EXACT_F32_FAMILIES):
        return False
    raise KeyError(f"matrix kernel {kernel_name!r} is in neither SPLIT_PRECISION_FAMILIES nor EXACT_F32_FAMILIES (villandiffusion_amd/flops.py)")
