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


def test_bench_quotes_pmc_traffic_only_from_a_profile_of_the_tree_kernels(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed PMC profile ONLY while the profile records the sha256 of
    the kernel source it was taken on and that equals the tree's; otherwise null with the reason."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    csrc = tmp_path / "gaunegf_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "k_chain1d_rs.hip").write_text("// kernel v2\n")
    sha = bench.kernel_source_id()
    assert sha is not None and len(sha) == 16
    name = "void chain1d_rs_kernel<51, 3, true>(...)"
    stale = {name: {"FETCH_SIZE": 10.0, "WRITE_SIZE": 20.0}, "_kernel_source_sha16": {"k_chain1d_rs.hip": "0" * 16}}
    (prof / "r09_pmc_c3_per_launch_avg.json").write_text(json.dumps(stale))
    val, why = bench.pmc_traffic_bytes("chain1d_rs_kernel")
    assert val is None and "another version" in why
    good = dict(stale, _kernel_source_sha16={"k_chain1d_rs.hip": sha})
    (prof / "r10_pmc_c3_per_launch_avg.json").write_text(json.dumps(good))
    val, src = bench.pmc_traffic_bytes("chain1d_rs_kernel")
    assert val == (2 * 10.0 + 20.0) * 1024.0 and src.endswith("r10_pmc_c3_per_launch_avg.json")
    # the committed round-3 profile belongs to the committed kernel
    monkeypatch.undo()
    val, src = bench.pmc_traffic_bytes("chain1d_rs_kernel")
    assert val is None or src.startswith("profiles/")
