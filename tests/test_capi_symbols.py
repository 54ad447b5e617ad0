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


def test_comm_rendezvous_rejects_a_stale_id_file(built, tmp_path, monkeypatch):
    """b2x_comm_init_session: a rank other than 0 accepts only a complete id file that carries the session's nonce.  The
    file a crashed earlier run left at the same path (another nonce, or no header at all) is rejected with an error that
    says so — it is never handed to ncclCommInitRank.  (No device needed: the wait ends before RCCL is touched.)"""
    import struct

    monkeypatch.setenv("B2X_COMM_TIMEOUT_S", "1")
    stale = tmp_path / "rccl_id"
    stale.write_bytes(b"B2XID001" + struct.pack("<Q", 1111) + bytes(128))  # an earlier session's id
    with pytest.raises(capi.B2XError, match="stale id rejected"):
        capi.Comm(1, 2, id_file=str(stale), nonce=2222)
    assert stale.exists()  # only rank 0 ever removes or replaces the file
    stale.write_bytes(bytes(128))  # the headerless format of round 2: rejected even when no nonce is asked for
    with pytest.raises(capi.B2XError, match="stale id rejected"):
        capi.Comm(1, 2, id_file=str(stale))
    with pytest.raises(capi.B2XError, match="no id of this session"):
        capi.Comm(1, 2, id_file=str(tmp_path / "never_written"), nonce=5)
