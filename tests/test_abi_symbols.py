"""CPU: the C-ABI library builds, loads, and exports every symbol include/rvc_hip.h declares (no compute without a GPU)."""
import os
import re

from conftest import ROOT


def _declared_symbols(experiments=False):
    """Symbols of the product ABI (experiments=False) or of the `#ifdef RVC_EXPERIMENTS` section at the end of the header."""
    hdr = open(os.path.join(ROOT, "include", "rvc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    m = re.search(r"#ifdef RVC_EXPERIMENTS(.*?)#endif", hdr, flags=re.S)
    assert m, "the header keeps its instrumentation hooks in an RVC_EXPERIMENTS section"
    part = m.group(1) if experiments else hdr[:m.start()] + hdr[m.end():]
    return sorted(set(re.findall(r"\b(rvc_[a-z0-9_]+)\s*\(", part)))


def test_library_exports_every_declared_symbol():
    from comfy_rvc_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in include/rvc_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == set(syms)
    exp = _declared_symbols(experiments=True)
    assert set(_lib.EXPERIMENT_SIGNATURES) == set(exp) and all(e.startswith("rvc_debug_") for e in exp)
    assert not any(s.startswith("rvc_debug_") for s in syms)


def test_product_library_has_no_experiment_hooks_and_only_documented_knobs():
    """The product build (csrc/Makefile, no -DRVC_EXPERIMENTS) exports no rvc_debug_* symbol and carries no environment-variable name beyond the documented
    knobs of INTEGRATION.md ("Run-time knobs"): experiment switches are compiled to their defaults and their names never reach the binary."""
    import subprocess
    from comfy_rvc_amd import _lib
    if os.environ.get("RVC_HIP_LIB"):
        return                                                       # a variant build is loaded on purpose
    assert not _lib.has_experiments
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(n.decode() for n in re.findall(rb"RVC_[A-Z0-9_]{2,}", blob))
    names = {n for n in names if not n.startswith(("RVC_HIP_CHECK", "RVC_REQUIRE", "RVC_TRY", "RVC_CATCH"))}
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## Run-time knobs"):]
    sec = sec[:sec.index("\n## ", 5)] if "\n## " in sec[5:] else sec
    documented = set(re.findall(r"`(RVC_[A-Z0-9_]+)`", sec))
    assert names <= documented, f"undocumented RVC_ strings in the product library: {sorted(names - documented)}"
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rvc_debug_" not in exported


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


def test_pair_arithmetic_knob():
    """RVC_H2 is the documented environment form of rvc_set_pair_arithmetic's initial mode (no GPU needed: the getter touches no device)."""
    import subprocess
    import sys
    code = "from comfy_rvc_amd import _lib; a = _lib.lib.rvc_get_pair_arithmetic(); _lib.check(_lib.lib.rvc_set_pair_arithmetic(1 - a)); print(a, _lib.lib.rvc_get_pair_arithmetic())"
    for env, want in (({}, "1 0"), ({"RVC_H2": "0"}, "0 1"), ({"RVC_H2": "1"}, "1 0")):
        e = dict(os.environ); e.pop("RVC_H2", None); e.update(env)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=e)
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == want, (env, out.stdout, out.stderr[-500:])
