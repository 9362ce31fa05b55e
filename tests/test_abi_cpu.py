"""CPU: the C-ABI shared library loads and exports every symbol include/vsx.h declares (no compute calls
without a GPU), the ctypes binding covers exactly those symbols, and the product path fails loudly on CPU."""

import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "vsx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vsx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from viscy_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `python -m viscy_amd.build`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vsx.h but not exported by libvsx.so"
    assert sorted(_lib.exported_symbols()) == syms  # binding and header agree
    assert _lib.lib().vsx_version() >= 1
    assert _lib.lib().vsx_get_flag(b"tn_tr") == 1
    # streaming (non-temporal) accesses are ON since round 3: their stores are compiler builtins, not the inline asm of round 1
    # (DESIGN §3 item 8); the soak / determinism tests run with the shipped values
    if not os.environ.get("VSX_FLAGS"):
        assert [_lib.lib().vsx_get_flag(f) for f in (b"nt_stream", b"grn_stream", b"ln_stream")] == [3, 2, 3]
    src = open(os.path.join(ROOT, "viscy_amd", "csrc", "vsx_common.h")).read()
    assert "global_store_dwordx4" not in src.split("__device__ __forceinline__ void stvec_stream")[1].split("ldvec_stream")[0]
    assert _lib.lib().vsx_set_flag(b"nope", 1) != 0
    assert b"unknown flag" in _lib.lib().vsx_last_error()


def test_struct_layout_matches_header():
    """VsxGemm is passed by pointer: field count / order of the ctypes mirror must track the header."""
    from viscy_amd import _lib

    src = open(os.path.join(ROOT, "include", "vsx.h")).read()
    body = src[src.index("typedef struct VsxGemm {") + len("typedef struct VsxGemm {") : src.index("} VsxGemm;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        for part in decl.split(","):
            m = re.search(r"(\w+)\s*(\[\d+\])?\s*$", part.strip())
            names.append(m.group(1))
    assert names == [f[0] for f in _lib.VsxGemm._fields_]


def test_no_cpu_fallback():
    from viscy_amd import ops
    from viscy_amd.losses import MixedLoss
    from viscy_amd.unext2 import UNeXt2

    with pytest.raises(RuntimeError, match="no CPU"):
        UNeXt2(backbone="convnextv2_atto")(torch.zeros(1, 1, 5, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU"):
        MixedLoss()(torch.zeros(1, 1, 5, 192, 192), torch.zeros(1, 1, 5, 192, 192))
    with pytest.raises(RuntimeError, match="not on a HIP device"):
        ops.ln_fwd(torch.zeros(4, 8), None, None, 4, 8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "viscy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", txt, flags=re.M), f
