"""No-GPU checks of the drop-in boundary: the shared library exists, exports every symbol the header
declares, and fails loudly (never falls back to a CPU path) when no device is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "nextgp_hip.h")).read()
    return sorted(set(re.findall(r"\b(ngp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(ngp):
    import __graft_entry__ as g
    g.build()
    lib = ngp.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nextgp_hip.h but not exported"
    assert sorted(ngp.SYMBOLS) == syms
    lib.ngp_abi_version.restype = C.c_int32
    assert lib.ngp_abi_version() == 2


def test_no_cpu_fallback(ngp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device failure mode cannot be exercised")
    with pytest.raises(ngp.NextGPHipError, match="no CPU fallback"):
        ngp.Sampler(device=0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under nextgp.jl_amd/ may reference it."""
    pkg = os.path.join(ROOT, "nextgp.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".jl")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "libngp_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_tools_and_bench_compile():
    """Every script that is sent to the GPU box parses on this interpreter (a syntax error there costs a GPU call)."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)
