"""The C-ABI shared library loads and exports every symbol include/boundmpc_hip.h declares (no compute calls).
Also: without a GPU the product refuses to construct a solver -- there is no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "boundmpc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bmpc_[a-z_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from boundmpc_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/boundmpc_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_options_and_error_strings():
    from boundmpc_amd import _lib
    lib = _lib.load()
    o = _lib.Options()
    assert lib.bmpc_default_options(ctypes.byref(o)) == 0
    assert o.tol == 1e-8 and o.max_iter == 500 and o.exact_hessian == 1
    assert lib.bmpc_error_string(0) == b"ok" and lib.bmpc_error_string(4) == b"no HIP device available"
    h = ctypes.c_void_p()
    assert lib.bmpc_create(0, 4, 0.1, None, ctypes.byref(h)) == 1      # invalid N
    assert lib.bmpc_create(10, 9, 0.1, None, ctypes.byref(h)) == 1     # invalid S


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from boundmpc_amd import BatchedOCPSolver, BoundMPCHipError
    with pytest.raises(BoundMPCHipError):
        BatchedOCPSolver(10, 4, 0.1)


def test_product_does_not_reference_oracle_or_emulator():
    """The product package must never import, link or execute anything under oracle/ or tests/."""
    pkg = os.path.join(ROOT, "boundmpc_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".inl", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                for pat in (r"^\s*(from|import)\s+oracle", r"^\s*(from|import)\s+tests", r"libbmpc_oracle", r"libbmpc_emu"):
                    assert not re.search(pat, src, flags=re.M), f"{f} refers to test infrastructure ({pat})"
