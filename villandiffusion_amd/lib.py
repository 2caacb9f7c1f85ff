"""ctypes binding of the gfx950 C-ABI library (include/villan_hip.h).

The product path has NO fallback: if ``libvillan_hip.so`` is missing, or a compute entry point is called
without a gfx950 device, this raises.  (Loading the library and listing its symbols works on a CPU-only
box -- that is what the ``-m "not gpu"`` tests check.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvillan_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# enums (villan_hip.h)
A_ROW, A_COL = 0, 1
B_PLAIN, B_KCONTIG, B_CONV3, B_CONV3_T, B_CONV3_S2, B_CONV3_UP, B_CONV3_DIL, B_CONVG = range(8)

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("D", _vp), ("bias", _vp), ("rowadd", _vp), ("residual", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32), ("a_mode", _i32), ("b_mode", _i32), ("NP", _i32),
                ("C", _i32), ("H", _i32), ("W", _i32), ("OH", _i32), ("OW", _i32),
                ("bias_on_n", _i32), ("d_trans", _i32), ("accumulate", _i32), ("tile", _i32), ("debug", _i32), ("alpha", _f32),
                ("lda", _i64), ("a_bstride", _i64), ("ldb", _i64), ("b_bstride", _i64),
                ("ldd", _i64), ("d_bstride", _i64), ("res_bstride", _i64), ("rowadd_bstride", _i64), ("ws", _vp), ("pad", _i32), ("nb2", _i32),
                ("a_b2stride", _i64), ("b_b2stride", _i64), ("d_b2stride", _i64), ("gn_ss", _vp), ("a_packed", _vp), ("a_packed_mpad", _i32), ("math", _i32), ("pool2", _i32),
                ("kh", _i32), ("kw", _i32), ("conv_stride", _i32), ("pad_h", _i32), ("pad_w", _i32), ("act", _i32),
                ("gn_part", _vp), ("act_out", _vp), ("act_bstride", _i64), ("b_presplit", _i32), ("reserved_", _i32)]


class WgradDesc(C.Structure):
    _fields_ = [("dY", _vp), ("X", _vp), ("dW", _vp), ("ws", _vp),
                ("M", _i32), ("C", _i32), ("T", _i32), ("nb", _i32), ("NP", _i32),
                ("H", _i32), ("W", _i32), ("OH", _i32), ("OW", _i32),
                ("mode", _i32), ("splits", _i32), ("accumulate", _i32), ("tile", _i32),
                ("dy_bstride", _i64), ("x_bstride", _i64), ("pad", _i32), ("math", _i32), ("presplit", _i32), ("reserved_", _i32)]


# name -> (restype, argtypes); every symbol declared in include/villan_hip.h
PROTOTYPES = {
    "vd_abi_version": (_i32, []),
    "vd_last_error": (C.c_char_p, []),
    "vd_async_errors": (_i32, [_i32]),
    "vd_device_ok": (_i32, []),
    "vd_gemm": (_i32, [C.POINTER(GemmDesc), _vp]),
    "vd_gemm_tile": (_i32, [C.POINTER(GemmDesc)]),
    "vd_gemm_ws_floats": (_i64, [C.POINTER(GemmDesc)]),
    "vd_conv_wgrad": (_i32, [C.POINTER(WgradDesc), _vp]),
    "vd_conv_wgrad_plan": (_i32, [C.POINTER(WgradDesc), C.POINTER(_i32), C.POINTER(_i32)]),
    "vd_conv_wgrad_ws_floats": (_i64, [C.POINTER(WgradDesc)]),
    "vd_presplit_pack": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_presplit_unpack": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_groupnorm_fwd_presplit_ok": (_i32, [_i32, _i32, _i32]),
    "vd_groupnorm_fwd_presplit": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i64, _i64, _vp]),
    "vd_groupnorm_bwd_presplit": (_i32, [_vp] * 12 + [_vp, _i32, _i32, _i32, _i32, _i32] + [_i64] * 7 + [_vp]),
    "vd_conv_wgrad_group_class": (_i32, [C.POINTER(WgradDesc)]),
    "vd_conv_wgrad_group_job_bytes": (_i64, []),
    "vd_conv_wgrad_group_variant": (_i32, [_i32]),
    "vd_conv_wgrad_group_plan": (_i32, [C.POINTER(WgradDesc), _i32, _vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)]),
    "vd_conv_wgrad_group_rebase": (_i32, [_vp, _i32, _vp, _vp]),
    "vd_conv_wgrad_group_launch": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp]),
    "vd_conv3_packed_bytes": (_i64, [_i32, _i32, _i32]),
    "vd_conv3_pack_weights": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_conv3_pack_weights_multi": (_i32, [_vp, _i32, _i64, _vp]),
    "vd_conv3_pack_weights_f16_multi": (_i32, [_vp, _i32, _i64, _vp]),
    "vd_weight_transpose": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_sumpool2x2": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _vp]),
    "vd_col2im_s2": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_rowsum": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_colsum": (_i32, [_vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "vd_colsum_segmented": (_i32, [_vp, _i32, _i32, _vp]),
    "vd_groupnorm_ws_floats": (_i64, [_i32, _i32, _i32, _i32]),
    "vd_groupnorm_stats": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i64, _vp]),
    "vd_groupnorm_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i64, _i64, _vp, _vp]),
    "vd_groupnorm_bwd": (_i32, [_vp] * 10 + [_i32] * 5 + [_i64] * 4 + [_vp, _vp]),
    "vd_groupnorm_stats_from_partials": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "vd_groupnorm_bwd_fused": (_i32, [_vp] * 12 + [_i32] * 5 + [_i64] * 6 + [_vp, _vp]),
    "vd_softmax_col_fwd": (_i32, [_vp, _i32, _i32, _vp]),
    "vd_softmax_col_bwd": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp]),
    "vd_attn_small_fwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i64, _i64, _vp]),
    "vd_attn_small_bwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i64, _i64, _i64, _vp]),
    "vd_attn_core_fwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "vd_attn_core_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "vd_attn_flash_fwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i64, _i64, _vp]),
    "vd_attn_flash_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i64, _i64, _i64, _i64, _vp]),
    "vd_timestep_embedding": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_silu_fwd": (_i32, [_vp, _vp, _i64, _vp]),
    "vd_silu_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "vd_add_strided": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp]),
    "vd_scale": (_i32, [_vp, _i64, _f32, _vp]),
    "vd_lincomb": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_f32), _i32, _i64, _vp]),
    "vd_qsample_backdoor": (_i32, [_vp] * 10 + [_i32, _i64, _vp]),
    "vd_mse_fwd_bwd": (_i32, [_vp] * 6 + [_i32, _i64, _f32, _vp]),
    "vd_loss_fwd_bwd": (_i32, [_vp] * 6 + [_i32, _i64, _f32, _i32, _vp]),
    "vd_l2norm_sq": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "vd_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "vd_sched_step": (_i32, [_vp] * 5 + [_i64] + [_f32] * 7 + [C.c_uint64, C.c_uint64, _vp]),
    "vd_batch_l2norm": (_i32, [_vp, _vp, _i32, _i64, _vp]),
    "vd_postprocess": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "vd_ssim": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    "vd_randn": (_i32, [_vp, _i64, C.c_uint64, C.c_uint64, _vp]),
    "vd_fir_resample2": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp]),
    "vd_fourier_embedding": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "vd_rowscale": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    "vd_vq_nearest": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_poison_batch": (_i32, [_vp] * 8 + [_i32] * 4 + [_f32, _f32, _i32, _vp]),
    "vd_pool3": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _vp]),
    "vd_resize_bilinear": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    "vd_channel_affine": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vd_lpips_layer": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
}

_lib: Optional[C.CDLL] = None


class VillanHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:], out.stderr[-4000:])
    if out.returncode != 0:
        raise VillanHipError("building libvillan_hip.so failed (see output above)")
    return LIB_PATH


def load() -> C.CDLL:
    """dlopen the library and bind every prototype; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: it ships its own HIP runtime (libamdhip64) and this library has to bind to that
    # SAME runtime instance -- device pointers and streams cross the boundary.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise VillanHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback exists). "
            f"Build it with `make -C {CSRC}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if lib.vd_abi_version() != 11:
        raise VillanHipError("libvillan_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error() -> str:
    return load().vd_last_error().decode("utf-8", "replace")


_SYNC_DEBUG = bool(os.environ.get("VD_SYNC_DEBUG"))     # debugging aid: synchronise after every launch and name it first


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise VillanHipError(f"{what} failed (rc={rc}): {last_error()}")
    if _SYNC_DEBUG:
        import sys
        import torch
        print(f"[vd] {what}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()


_device_checked = False


def require_device() -> None:
    """Called by every compute wrapper once: a gfx950 device must be present."""
    global _device_checked
    if _device_checked:
        return
    lib = load()
    rc = lib.vd_device_ok()
    if rc != 0:
        raise VillanHipError(f"MI355X (gfx950) device required: {last_error()}")
    _device_checked = True
