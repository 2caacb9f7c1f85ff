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
    assert handle.vd_abi_version() == 3


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
