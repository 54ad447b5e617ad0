"""The C-ABI library loads on a machine without a GPU and exports every symbol include/b2x.h declares;
compute entry points fail loudly (no CPU fallback) when no device is present."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from block2_preview_amd import capi
from block2_preview_amd.planfile import PAIR_DTYPE


def _declared_in_header():
    txt = open(os.path.join(ROOT, "include", "b2x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(b2x_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared_in_header() == sorted(capi.DECLARED_SYMBOLS)


def test_library_exports_every_declared_symbol(built):
    lib = capi.lib()
    for name in _declared_in_header():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.b2x_version()


def test_pair_struct_size_matches_numpy_dtype():
    txt = open(os.path.join(ROOT, "include", "b2x.h")).read()
    assert "typedef struct b2x_pair" in txt and PAIR_DTYPE.itemsize == 96


def test_no_cpu_fallback_without_device(built):
    if capi.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(capi.B2XError):
        capi.device_init(0)
    with pytest.raises(capi.B2XError):
        capi.Arena.from_host([np.zeros(8)])


def test_product_library_carries_no_host_emulator(built):
    """the host evaluation of compiled plans lives in tests/native/libb2x_testhooks.so, not in the shipped library"""
    lib = capi.lib()
    for name in ("b2x_debug_compile_and_emulate", "b2x_debug_compile_and_emulate_gemms",
                 "b2x_debug_compile_and_emulate_outer", "b2x_debug_compile_diag"):
        assert not hasattr(lib, name), name
