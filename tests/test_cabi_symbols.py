"""CPU-only: the C-ABI library builds, loads and exports every symbol the header declares."""
import ctypes
import re

import pytest

from tetrad_amd import _lib


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def header_symbols():
    text = _lib.HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tq_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} not exported by {_lib.LIB_PATH}"


def test_no_device_is_a_clean_error(lib):
    """Without a GPU tq_create must fail with an error code and a message, never crash."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = lib.tq_create(ctypes.byref(h), 0)
    assert rc == -2 and not h.value
    assert b"HIP device" in lib.tq_last_error(None)


def test_null_context_is_rejected(lib):
    assert lib.tq_set_option(None, b"nrep", 8) == -1
    assert lib.tq_timing_enable(None, 1) == -1
    lib.tq_destroy(None)  # no-op


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (test infrastructure)."""
    from pathlib import Path
    for p in Path(_lib.CSRC).parent.rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p
