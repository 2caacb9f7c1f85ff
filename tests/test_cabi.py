"""The C-ABI library loads on a CPU-only box and exports every symbol include/villan_hip.h declares; the ctypes
mirrors of the descriptor structs have the C layout; compute entry points fail loudly without a GPU."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from villandiffusion_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "villan_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    handle = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in villan_hip.h but not exported"
        assert s in lib.PROTOTYPES, f"{s} has no ctypes prototype in lib.py"
    assert set(lib.PROTOTYPES) == set(syms)
    assert handle.vd_abi_version() == 11


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "villan_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(vd_gemm_desc), offsetof(vd_gemm_desc, alpha), offsetof(vd_gemm_desc, lda), offsetof(vd_gemm_desc, rowadd_bstride),'
                   'sizeof(vd_wgrad_desc), offsetof(vd_wgrad_desc, dy_bstride));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    c = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    G, W = lib.GemmDesc, lib.WgradDesc
    assert c == [ctypes.sizeof(G), G.alpha.offset, G.lda.offset, G.rowadd_bstride.offset, ctypes.sizeof(W), W.dy_bstride.offset]


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    from villandiffusion_amd import ops
    with pytest.raises(lib.VillanHipError, match="gfx950|device"):
        ops.silu_fwd(torch.zeros(4), torch.zeros(4))
    # pure host queries still work
    assert ops.wgrad_ws_floats(128, 128, 9, 128, 1024) > 0


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.VillanHipError, match="no CPU fallback"):
        lib.load()


def test_split_precision_planner_agrees_with_the_python_side_eligibility():
    """vd_gemm_tile / vd_conv_wgrad_plan are pure host functions: the kernels the library would pick must match the predicates the
    host side (ops.*_eligible) uses to decide whether to hand over packed operands / math = 1 -- a mismatch would surface as
    VD_EINVAL at run time on some exotic layer shape.  No compute calls, no GPU."""
    import ctypes as C
    from villandiffusion_amd import lib, ops
    from villandiffusion_amd.lib import (A_COL, A_ROW, B_CONV3, B_CONV3_S2, B_CONV3_T, B_CONV3_UP, B_KCONTIG, B_PLAIN, GemmDesc,
                                         WgradDesc)
    h = lib.load()
    FAKE = 0x10000                                   # 16-byte aligned dummy addresses: the planners only look at alignment
    for mode in (B_CONV3, B_CONV3_T, B_CONV3_UP, B_CONV3_S2):
        for OW in (2, 4, 8, 16, 24, 32, 64, 96, 128, 256):
            for (M, Cc) in ((128, 128), (64, 16), (200, 384), (32, 128), (128, 24), (3, 128), (128, 3)):
                for nb in (1, 3, 128):
                    OH = OW
                    H = OH // 2 if mode == B_CONV3_UP else (2 * OH if mode == B_CONV3_S2 else OH)
                    if H < 1:
                        continue
                    d = GemmDesc()
                    d.A, d.B, d.D, d.a_packed = FAKE, FAKE, FAKE, FAKE
                    d.a_packed_mpad = (M + 127) // 128 * 128
                    d.M, d.N, d.K, d.NP = M, nb * OH * OW, Cc * 9, OH * OW
                    d.a_mode, d.b_mode = A_ROW, mode
                    d.C, d.H, d.W, d.OH, d.OW = Cc, H, H, OH, OW
                    d.lda, d.b_bstride, d.ldd, d.d_bstride, d.alpha = Cc * 9, Cc * H * H, OH * OW, M * OH * OW, 1.0
                    want = ops.bx3_eligible(M, Cc, OH, OW, mode)
                    got = h.vd_gemm_tile(C.byref(d))
                    # 17 / 18: the 16x16x32-MFMA kernels; 20: the whole-K kernel of the 8x8 / 4x4 levels (round 6)
                    assert (got in (8, 12, 15, 16, 17, 18, 20)) == want and got in (8, 12, 15, 16, 17, 18, 20, -1), (mode, OW, M, Cc, nb, got, want)
                    if got == 20:                                     # no split-K slabs; only the stride-1 kinds at 8x8 / 4x4 on grids that fill the chip
                        assert h.vd_gemm_ws_floats(C.byref(d)) == 0 and OW in (4, 8) and mode in (B_CONV3, B_CONV3_T) and Cc % 32 == 0, (mode, OW, M, Cc, nb)
                    if mode == B_CONV3_T and ops.bx3_pool2_eligible(M, Cc, OH, OW, nb):      # pool2 needs the unsplit grid
                        assert got in (8, 12, 15, 17, 18) and h.vd_gemm_ws_floats(C.byref(d)) == 0, (OW, M, Cc, nb)
                    if mode in (B_CONV3, B_CONV3_UP):
                        w = WgradDesc()
                        w.dY, w.X, w.dW = FAKE, FAKE, FAKE
                        w.M, w.C, w.T, w.nb, w.NP = M, Cc, 9, nb, OH * OW
                        w.H, w.W, w.OH, w.OW, w.mode, w.math = H, H, OH, OW, mode, 1
                        w.dy_bstride, w.x_bstride = M * OH * OW, Cc * H * H
                        tile, splits = C.c_int32(0), C.c_int32(0)
                        assert h.vd_conv_wgrad_plan(C.byref(w), C.byref(tile), C.byref(splits)) == 0
                        assert (tile.value == 4) == ops.wgrad_bx3_eligible(M, Cc, OH, OW, mode), (mode, OW, M, Cc, nb, tile.value)
    # non-square images: only the row-segment (wide) tiles take them
    for (OH, OW) in ((16, 64), (3, 128), (10, 256), (48, 96), (16, 32), (8, 16)):
        for mode in (B_CONV3, B_CONV3_T):
            d = GemmDesc()
            d.A, d.B, d.D, d.a_packed, d.a_packed_mpad = FAKE, FAKE, FAKE, FAKE, 128
            d.M, d.N, d.K, d.NP, d.a_mode, d.b_mode = 128, 2 * OH * OW, 64 * 9, OH * OW, A_ROW, mode
            d.C, d.H, d.W, d.OH, d.OW = 64, OH, OW, OH, OW
            d.lda, d.b_bstride, d.ldd, d.d_bstride, d.alpha = 64 * 9, 64 * OH * OW, OH * OW, 128 * OH * OW, 1.0
            assert (h.vd_gemm_tile(C.byref(d)) in (8, 12, 15, 16, 17, 18)) == ops.bx3_eligible(128, 64, OH, OW, mode), (OH, OW, mode)
        w = WgradDesc()
        w.dY, w.X, w.dW = FAKE, FAKE, FAKE
        w.M, w.C, w.T, w.nb, w.NP, w.H, w.W, w.OH, w.OW, w.mode, w.math = 128, 64, 9, 2, OH * OW, OH, OW, OH, OW, B_CONV3, 1
        w.dy_bstride, w.x_bstride = 128 * OH * OW, 64 * OH * OW
        tile, splits = C.c_int32(0), C.c_int32(0)
        h.vd_conv_wgrad_plan(C.byref(w), C.byref(tile), C.byref(splits))
        assert (tile.value == 4) == ops.wgrad_bx3_eligible(128, 64, OH, OW, B_CONV3), (OH, OW, tile.value)
    # 1x1 convolutions (shared packed A) and activation products (math = 1)
    for NP in (16, 64, 100, 256, 1024):
        for (M, K) in ((256, 512), (64, 16), (200, 80), (32, 256), (256, 24)):
            for nb in (1, 2, 8, 128):
                d = GemmDesc()
                d.A, d.B, d.D, d.a_packed = FAKE, FAKE, FAKE, FAKE
                d.a_packed_mpad = (M + 127) // 128 * 128
                d.M, d.N, d.K, d.NP = M, nb * NP, K, NP
                d.a_mode, d.b_mode = A_ROW, B_PLAIN
                d.lda, d.ldb, d.b_bstride, d.ldd, d.d_bstride, d.alpha = K, NP, K * NP, NP, M * NP, 1.0
                assert (h.vd_gemm_tile(C.byref(d)) in (9, 11, 13, 19)) == ops.gemm_bx3_eligible(M, K, NP, nb), (NP, M, K, nb)
                for a_mode, b_mode in ((A_COL, B_PLAIN), (A_ROW, B_PLAIN), (A_ROW, B_KCONTIG)):
                    e = GemmDesc()
                    e.A, e.B, e.D, e.math = FAKE, FAKE, FAKE, 1
                    e.M, e.N, e.K, e.NP = M, nb * NP, K, NP
                    e.a_mode, e.b_mode = a_mode, b_mode
                    e.lda = K if a_mode == A_ROW else M
                    e.ldb = K if b_mode == B_KCONTIG else NP
                    e.a_bstride, e.b_bstride, e.ldd, e.d_bstride, e.alpha = M * K, K * NP, NP, M * NP, 1.0
                    assert (h.vd_gemm_tile(C.byref(e)) == 10) == ops.gemm_bx3_act_eligible(M, K, NP), (NP, M, K, nb, a_mode, b_mode)
