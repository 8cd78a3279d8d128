"""CPU: the C-ABI library builds, loads, and exports every symbol include/rvc_hip.h declares (no compute without a GPU)."""
import os
import re

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "rvc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rvc_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from comfy_rvc_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in include/rvc_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == set(syms)


def test_no_cpu_fallback_without_gpu():
    import ctypes as C
    import torch
    from comfy_rvc_amd import _lib
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    assert _lib.lib.rvc_ctx_create(0, C.byref(h)) != 0
    assert b"device" in _lib.lib.rvc_last_error().lower()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "comfy-rvc_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)
