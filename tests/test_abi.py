"""CPU-only: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol that
include/pyani_gpu.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from pyani_amd import build, _lib
    build.build_gpu()
    return _lib.load()


def declared_symbols():
    text = (ROOT / "include" / "pyani_gpu.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from pyani_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 20
    assert set(syms) == set(_lib.SIGNATURES)
    for s in syms:
        assert hasattr(lib, s), f"libpyani_gpu.so does not export {s}"


def test_version_and_kernel_names(lib):
    assert b"gfx950" in lib.pg_version()
    assert lib.pg_kernel_name(0) == b"tetra_count_kernel"


def test_no_cpu_fallback(lib):
    """Without a HIP device pg_create must fail loudly; with one it must succeed."""
    import torch
    h = ctypes.c_void_p()
    rc = lib.pg_create(ctypes.byref(h), 0)
    if torch.cuda.is_available():
        assert rc == 0
        lib.pg_destroy(h)
    else:
        assert rc == -2 and not h.value
        from pyani_amd import _lib
        from pyani_amd.engine import Engine
        with pytest.raises(_lib.PyaniGpuError):
            Engine(0)


def test_product_never_touches_oracle():
    """The package must not import, link or execute anything under oracle/ (that would void parity claims)."""
    for py in (ROOT / "pyani_amd").rglob("*"):
        if py.suffix in {".py", ".cpp", ".hip", ".h", ".inc"}:
            txt = py.read_text()
            assert "oracle" not in txt.lower() or py.name == "build.py", py
    # build.py only COMPILES the checker; it must not load it
    assert "CDLL" not in (ROOT / "pyani_amd" / "build.py").read_text()
