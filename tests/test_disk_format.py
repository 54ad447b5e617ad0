"""On-disk format of SparseMatrix / SparseMatrixInfo (src/core/sparse_matrix.hpp:500-566, 896-971):
files WRITTEN BY THE REFERENCE (MPS tensors saved with SparseMatrix::save_data(file, true)) are read by the host mirror,
and the mirror's writer reproduces them byte for byte.  No GPU."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from block2_preview_amd.planfile import read_arrays

FILES = sorted(glob.glob(os.path.join(GOLDEN, "disk_*.tensor")))


def _sym(fn):
    return "su2" if "su2" in os.path.basename(fn) else "sz"


def test_fixtures_present():
    assert len(FILES) >= 3


@pytest.mark.parametrize("fn", FILES, ids=os.path.basename)
def test_load_reference_file(built, fn):
    from block2_preview_amd import b2x_host

    exp = read_arrays(fn + ".arr")
    got = b2x_host.sparse_matrix_load(_sym(fn), fn)
    iid = int(exp["info"][0])
    pre = "info.%d." % iid
    assert np.array_equal(np.array(got["quanta"], np.uint64), exp[pre + "quanta"])
    assert np.array_equal(np.array(got["nbra"], np.uint32), exp[pre + "nbra"])
    assert np.array_equal(np.array(got["nket"], np.uint32), exp[pre + "nket"])
    assert np.array_equal(np.array(got["ntot"], np.uint32), exp[pre + "ntot"])
    assert [int(x) for x in got["meta"][:3]] == [int(x) for x in exp[pre + "meta"][:3]]
    assert int(got["meta"][3]) == int(exp["info"][1]) == len(exp["data"])
    assert got["factor"] == exp["factor"][0]
    assert np.array_equal(got["data"], exp["data"])  # bit-exact


@pytest.mark.parametrize("fn", FILES, ids=os.path.basename)
def test_save_is_byte_identical(built, fn, tmp_path):
    from block2_preview_amd import b2x_host

    got = b2x_host.sparse_matrix_load(_sym(fn), fn)
    out = str(tmp_path / "copy.tensor")
    b2x_host.sparse_matrix_save(_sym(fn), out, got["quanta"], got["nbra"], got["nket"], got["ntot"], got["meta"],
                                got["factor"], got["data"])
    assert open(out, "rb").read() == open(fn, "rb").read()


def test_refuses_bad_files(built, tmp_path):
    from block2_preview_amd import b2x_host

    with pytest.raises(RuntimeError):
        b2x_host.sparse_matrix_load("su2", str(tmp_path / "missing.tensor"))
    bad = tmp_path / "short.tensor"
    bad.write_bytes(open(FILES[0], "rb").read()[:40])
    with pytest.raises(RuntimeError):
        b2x_host.sparse_matrix_load(_sym(FILES[0]), str(bad))
