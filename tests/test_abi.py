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
    assert lib.ngp_abi_version() == 4


def test_no_cpu_fallback(ngp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device failure mode cannot be exercised")
    with pytest.raises(ngp.NextGPHipError, match="no CPU fallback"):
        ngp.Sampler(device=0)


def test_exception_barrier_at_the_c_abi(ngp):
    """include/nextgp_hip.h: "No C++ exception crosses this boundary".  Entry points run inside one try / catch (NGP_TRY / NGP_CATCH
    in csrc/ngp_api.hip); a throw inside becomes a negative status and a message, not an abort of the calling process (a Julia
    ccall frame cannot be unwound).  Runs without a GPU: these entry points need no handle."""
    lib = ngp.load()
    lib.ngp_debug_throw.restype = C.c_int32
    lib.ngp_write_panel_file.restype = C.c_int32
    # (1) a real allocation failure inside a real entry point: a two-bit column buffer of 2^60 bytes -> std::bad_alloc / length_error
    g = (C.c_uint8 * 4)(0, 1, 2, 0)
    path = os.path.join(ROOT, "gpurun_out", "never_written.ngp")
    rc = lib.ngp_write_panel_file(path.encode(), g, C.c_int64(1 << 62), C.c_int64(1), C.c_int64(1 << 62), C.c_int32(2))
    assert rc in (-4, -3) and not os.path.exists(path)
    msg = lib.ngp_last_error(None).decode()
    assert "memory" in msg or "internal error" in msg
    # (2) the three catch clauses, through the test hook that throws past the same barrier
    assert lib.ngp_debug_throw(None, C.c_int32(0)) == -4 and "bad_alloc" in lib.ngp_last_error(None).decode()
    assert lib.ngp_debug_throw(None, C.c_int32(1)) == -3 and "C++ exception stopped at the C ABI" in lib.ngp_last_error(None).decode()
    assert lib.ngp_debug_throw(None, C.c_int32(2)) == -3 and "unknown C++ exception" in lib.ngp_last_error(None).decode()
    assert lib.ngp_debug_throw(None, C.c_int32(7)) == -1
    # (3) every extern "C" body in the source sits inside the barrier (a new entry point without it fails here)
    src = open(os.path.join(ROOT, "nextgp.jl_amd", "csrc", "ngp_api.hip")).read()
    unguarded, defined = [], {"ngp_last_error"}   # (ngp_last_error returns a stored C string, ngp_abi_version a constant)
    lines = src.split("\n")
    for i, line in enumerate(lines):
        m = re.match(r"^int32_t (ngp_[a-z0-9_]+)\(", line)
        if not m:
            continue
        defined.add(m.group(1))
        if line.rstrip().endswith("}"):       # one-line body (ngp_abi_version)
            continue
        end = next(k for k in range(i, len(lines)) if lines[k] == "}")
        if not any("NGP_TRY" in b for b in lines[i:end]) or not any("NGP_CATCH" in b for b in lines[i:end]):
            unguarded.append(m.group(1))
    assert sorted(defined) == header_symbols() and not unguarded, (unguarded, sorted(set(header_symbols()) ^ defined))


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
