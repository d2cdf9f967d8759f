"""The C-ABI library: builds, loads, exports every symbol include/negf.h declares,
and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "negf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(negf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gaunegf_amd import _lib, build
    build.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/negf.h but not exported"
    # the ctypes table binds exactly the header's symbols
    assert sorted(_lib.SIGNATURES) == names


def test_error_strings_and_version():
    from gaunegf_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.negf_version()
    assert lib.negf_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.negf_strerror(_lib.NEGF_ENODEV)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    from gaunegf_amd import _lib
    lib = _lib.load()
    if lib.negf_device_count() > 0:
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    assert lib.negf_create(ctypes.byref(ctx), 0) == _lib.NEGF_ENODEV
    from gaunegf_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine()
    import numpy as np
    from gaunegf_amd.integrate import GrInt
    from gaunegf_amd.surfGTester import surfGTest
    F = np.zeros((4, 4)); S = np.eye(4)
    g = surfGTest(F, S, [[0], [3]], -0.1j)
    with pytest.raises(RuntimeError):
        GrInt(F, S, g, np.array([0.1]), np.array([1.0]))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under gaunegf_amd/ (nor bench.py's GPU
    path) may reference it."""
    pkg = os.path.join(ROOT, "gaunegf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
