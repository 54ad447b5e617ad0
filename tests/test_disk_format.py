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


# ---- compressed storage (FPCodec, src/core/fp_codec.hpp:158-; SparseMatrix::save_data with
# frame->compressed_sparse_tensor_storage, src/core/sparse_matrix.hpp:937-957) -----------------------------------------
FPC = sorted(glob.glob(os.path.join(GOLDEN, "diskc_*.tensor.fpc")))
FPC_ARGS = {"diskc_n2su2": (1e-8, 64), "diskc_h10sz": (1e-5, 1024)}  # (precision, chunk) the reference wrote them with


def _args(fn):
    return FPC_ARGS[os.path.basename(fn).split(".")[0]]


def test_compressed_fixtures_present():
    assert len(FPC) >= 2


@pytest.mark.parametrize("fn", FPC, ids=os.path.basename)
def test_compressed_file_written_by_the_reference(built, fn, tmp_path):
    """the reference wrote the same MPS tensor plain (.tensor) and through its FPCodec (.tensor.fpc): the mirror decodes
    the coded file to the plain values within the precision, and coding the plain values reproduces the reference's coded
    file byte for byte (the codec is a bit stream: parity is exact)"""
    from block2_preview_amd import b2x_host

    prec, chunk = _args(fn)
    raw = b2x_host.sparse_matrix_load(_sym(fn), fn[:-4])
    dec = b2x_host.sparse_matrix_load(_sym(fn), fn)
    assert dec["quanta"] == raw["quanta"] and dec["factor"] == raw["factor"] and len(dec["data"]) == len(raw["data"])
    err = np.abs(np.asarray(dec["data"]) - np.asarray(raw["data"]))
    assert 0 < err.max() < 2 * prec  # lossy: what lies below 2^(exponent of prec + 1) may be dropped, as in the reference
    out = str(tmp_path / "re.fpc")
    b2x_host.sparse_matrix_save(_sym(fn), out, raw["quanta"], raw["nbra"], raw["nket"], raw["ntot"], raw["meta"],
                                raw["factor"], raw["data"], fp_prec=prec, fp_chunk=chunk)
    assert open(out, "rb").read() == open(fn, "rb").read()
    # decoded values survive another pass through the codec unchanged
    out2 = str(tmp_path / "re2.fpc")
    b2x_host.sparse_matrix_save(_sym(fn), out2, dec["quanta"], dec["nbra"], dec["nket"], dec["ntot"], dec["meta"],
                                dec["factor"], dec["data"], fp_prec=prec, fp_chunk=chunk)
    assert np.array_equal(b2x_host.sparse_matrix_load(_sym(fn), out2)["data"], dec["data"])


@pytest.mark.parametrize("prec,chunk,n", [(1e-6, 7, 100), (1e-12, 4096, 5000), (0.5, 16, 33), (1e-3, 1, 3), (1e-9, 64, 0)])
def test_codec_round_trip_properties(built, prec, chunk, n):
    """random arrays over many magnitudes (zeros, tiny values, both signs): |x - decode(encode(x))| < 2 prec (a value whose
    exponent equals that of prec decodes to zero when it is the smallest of its chunk, exactly as in the reference's
    decoder, fp_codec.hpp:204-206); larger values keep their sign; the coded stream never exceeds one word per element
    plus one per chunk; decoded values pass through the codec unchanged"""
    from block2_preview_amd import b2x_host

    rng = np.random.default_rng(int(chunk) + n)
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-15, 6, n)
    if n > 4:
        x[::5] = 0.0
    blob = b2x_host.fpcodec_encode(x, prec, chunk)
    y = np.asarray(b2x_host.fpcodec_decode(blob, n))
    assert len(y) == n
    if n:
        assert np.abs(y - x).max() < 2 * prec
        big = np.abs(x) > 4 * prec
        assert np.array_equal(np.sign(y[big]), np.sign(x[big]))
    n_chunks = (n + chunk - 1) // chunk
    assert len(blob) <= 4 + 8 + 8 * n_chunks + 8 * (n + n_chunks) + 4
    assert np.array_equal(np.asarray(b2x_host.fpcodec_decode(b2x_host.fpcodec_encode(y, prec, chunk), n)), y)


def test_partition_file_names(built):
    """scratch file names of MovingEnvironment / MPS (src/dmrg/moving_environment.hpp:857-880, src/dmrg/mps.hpp): the names
    the reference created in its scratch directory during the run that wrote the fixtures"""
    from block2_preview_amd.planfile import partition_filename, mps_tensor_filename

    assert partition_filename("/tmp/s", "F0", "DMRG", True, 3) == "/tmp/s/F0.PART.DMRG.LEFT.3"
    assert partition_filename("/tmp/s", "F0", "DMRG", False, 12, info=True) == "/tmp/s/F0.PART.INFO.DMRG.RIGHT.12"
    assert mps_tensor_filename("/tmp/s", "F", "KET", 5) == "/tmp/s/F.MPS.KET.5"
    listed = open(os.path.join(GOLDEN, "scratch_listing.txt")).read().split()
    made = {os.path.basename(partition_filename("/x", "F0", "DMRG", left, i, info)) for left in (True, False)
            for i in range(10) for info in (True, False)}
    made |= {os.path.basename(mps_tensor_filename("/x", "F", "KET", i)) for i in range(-1, 10)}
    parts = [f for f in listed if ".PART." in f or (f.startswith("F.MPS.KET.") )]
    assert parts and all(f in made for f in parts), [f for f in parts if f not in made]


PART = sorted(glob.glob(os.path.join(GOLDEN, "part_n2su2", "F0.PART.DMRG.RIGHT.*")))


def test_partition_file_content_round_trip(built, tmp_path):
    """partition-file CONTENT (MovingEnvironment: frame_->save_data(1, get_right_partition_filename(i)),
    src/dmrg/moving_environment.hpp:428-440; DataFrame::save_data_to, src/core/allocator.hpp:580-592): the files the reference
    left in its scratch directory after building the initial environments of N2/STO-3G SU2 M=200 with its default stack
    allocation (operators on the frame stack; the run stops at the first site) are read into their two stacks and written
    back byte for byte; the double stack of RIGHT.i is the renormalised right block of sites i+2.. — its length is the
    length of that block in the event chain of the same run (tests/golden/part_n2su2/n2p.zip)"""
    from block2_preview_amd.planfile import read_arrays, read_partition_file, write_partition_file
    from block2_preview_amd.sweep import ChainFixture

    assert len(PART) == 8
    fx = ChainFixture(os.path.join(GOLDEN, "part_n2su2", "n2p"))
    last_rrot = set()  # lengths of the right blocks: rotation (+ the operator sums the NC -> CN transform events that follow
    for pos, (n, kind, fn) in enumerate(fx.events):  # it form inside the rotated block)
        if kind == "rrot":
            total = int(read_arrays(fn)["meta"][6])
            for n2, k2, fn2 in fx.events[pos + 1:]:
                if k2 not in ("rntr", "rint"):
                    break
                total = int(read_arrays(fn2)["meta"][3])
            last_rrot.add(total)
    matched = 0
    for fn in PART:
        ist, dst = read_partition_file(fn)
        out = str(tmp_path / os.path.basename(fn))
        write_partition_file(out, ist, dst)
        assert open(out, "rb").read() == open(fn, "rb").read()
        matched += len(dst) in last_rrot
        assert np.isfinite(dst).all() and np.abs(dst).max() > 0
    # (7 of the 8 blocks have exactly the chain's length; the block at the NC -> CN switch of the MPO holds, in the
    #  reference's stack, only the operators the next blocking reads: 7 360 elements against the chain's 9 244)
    assert matched >= 7
    # compressed storage of the same stacks: coded and decoded through the FPCodec mirror within its precision
    ist, dst = read_partition_file(PART[2])
    out = str(tmp_path / "coded")
    write_partition_file(out, ist, dst, fp_prec=1e-8, fp_chunk=64)
    ist2, dst2 = read_partition_file(out, fp_prec=1e-8, fp_chunk=64)
    assert np.array_equal(ist2, ist) and np.abs(dst2 - dst).max() < 2e-8 and os.path.getsize(out) < os.path.getsize(PART[2])
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad"), "wb").write(open(PART[0], "rb").read()[:-8])
        read_partition_file(str(tmp_path / "bad"))
