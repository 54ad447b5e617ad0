"""Blocking on the MI355X path (b2x_outer_build through the C ABI / the host mirror) vs the enlarged operators the real
reference computed and vs the oracle.  fp64, 1e-12 relative to max|result|; repeat runs are bitwise identical."""
import os

import numpy as np
import pytest

from oracle import oracle
from test_blocking import BLK, load_blk, numpy_outer, random_outer_terms

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-12):
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("fn", BLK, ids=os.path.basename)
def test_golden_reference_blocking(gpu, fn):
    t, d = load_blk(fn)
    arena = gpu.Arena.from_host([d["arena"]])
    out = np.zeros(int(d["lens"][3]))
    gpu.outer_build(arena, t, d["in"], out)
    again = np.zeros_like(out)
    gpu.outer_build(arena, t, d["in"], again)
    arena.close()
    assert _close(out, d["out_ref"])
    assert out.tobytes() == again.tobytes()


@pytest.mark.parametrize("seed", range(5))
def test_random_terms_vs_oracle(gpu, seed):
    rng = np.random.default_rng(60 + seed)
    t, in_len, out_len, arena_len = random_outer_terms(rng, 600, max_dim=[6, 25, 80, 200, 400][seed])
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    out0 = rng.standard_normal(out_len)
    ref = out0.copy()
    oracle.outer(t, arena, vin, ref)
    ar = gpu.Arena.from_host([arena])
    out = out0.copy()
    gpu.outer_build(ar, t, vin, out)
    ar.close()
    assert _close(out, ref)


@pytest.mark.parametrize("seed", range(3))
def test_two_step_outer_plan_equals_outer_build(gpu, seed):
    """b2x_outer_plan_create / _execute / _destroy (a list compiled before its data exist, executed on device vectors later,
    twice over different data) against the oracle and bitwise against the one-call form"""
    rng = np.random.default_rng(160 + seed)
    t, in_len, out_len, arena_len = random_outer_terms(rng, 500, max_dim=[9, 60, 250][seed])
    op = gpu.OuterPlan(t, arena_len, in_len, out_len)
    for rep in range(2):
        arena, vin, out0 = rng.standard_normal(arena_len), rng.standard_normal(in_len), rng.standard_normal(out_len)
        ref = out0.copy()
        oracle.outer(t, arena, vin, ref)
        ar = gpu.Arena.from_host([arena])
        d_in, d_out = gpu.DeviceBuffer(in_len, vin), gpu.DeviceBuffer(out_len, out0)
        op.execute(ar, d_in.ptr, d_out.ptr)
        gpu.device_sync()
        got = d_out.download()
        one = out0.copy()
        gpu.outer_build(ar, t, vin, one)
        assert _close(got, ref) and got.tobytes() == one.tobytes()
        wrong = gpu.Arena.from_host([np.zeros(arena_len + 1)])
        with pytest.raises(gpu.B2XError):
            op.execute(wrong, d_in.ptr, d_out.ptr)
        wrong.close(), ar.close()
    op.close()


def test_host_mirror_outer_perform(gpu):
    """C++ host mirror: load the reference's terms, BatchGEMMSeq::outer_perform == reference result"""
    from block2_preview_amd import b2x_host

    t, d = load_blk(BLK[0])
    seq = b2x_host.BatchGEMMSeq()
    out = np.zeros(int(d["lens"][3]))
    seq.load_outer(t, d["arena"], d["in"], out)
    assert seq.n_outer == len(t)
    seq.outer_perform(out)
    assert seq.n_outer == 0
    assert _close(out, d["out_ref"])


def test_kron_and_iadd_semantics(gpu):
    """tensor_product / iadd of the host mirror vs numpy.kron (general, transposed, scalar cases), in the manner of the
    reference's tensor_product test (unit_test/test_matrix.cpp)"""
    from block2_preview_amd import b2x_host

    rng = np.random.default_rng(77)
    for conja in (False, True):
        for conjb in (False, True):
            for shape_a, shape_b in (((3, 4), (2, 5)), ((1, 1), (4, 3)), ((5, 2), (1, 1))):
                a, b = rng.random(shape_a), rng.random(shape_b)
                oa, ob = (a.T if conja else a), (b.T if conjb else b)
                k = np.kron(oa, ob)
                rows, cols = k.shape[0] + 3, k.shape[1] + 4
                out = rng.random(rows * cols)
                ref = out.reshape(rows, cols).copy()
                ref[2:2 + k.shape[0], 1:1 + k.shape[1]] += 0.7 * k
                seq = b2x_host.BatchGEMMSeq()
                seq.tensor_product(a, conja, b, conjb, (out, 0, rows, cols), 0.7, 2 * cols + 1)
                seq.outer_perform(out)
                assert _close(out.reshape(rows, cols), ref), (conja, conjb, shape_a, shape_b)
    x, y = rng.random((4, 6)), rng.random(24)
    seq = b2x_host.BatchGEMMSeq()
    seq.iadd((y, 0, 4, 6), x, -0.5)
    y0 = y.copy()
    seq.outer_perform(y)
    assert _close(y, y0 + (-0.5 * x).ravel())
    z = rng.random(24)
    z0 = z.copy()
    seq.iadd((z, 0, 6, 4), x, 2.0, True)
    seq.outer_perform(z)
    assert _close(z, z0 + (2.0 * x.T).ravel())


def test_cr2_blocking_structure(gpu):
    """Cr2/SVP M=250 blocking (31k terms, structure recorded by the reference), synthetic data: vs oracle"""
    import glob

    from conftest import GOLDEN
    from block2_preview_amd.planfile import read_outer_struct_npz

    fn = sorted(glob.glob(os.path.join(GOLDEN, "*sw1_c20_rblk.blkstruct.npz")))[0]
    t, lens = read_outer_struct_npz(fn)
    in_len, out_len = int(lens[2]), int(lens[3])
    rng = np.random.default_rng(8)
    arena, vin = rng.random(int(lens[1])) - 0.5, rng.random(in_len) - 0.5
    ref = np.zeros(out_len)
    oracle.outer(t, arena, vin, ref)
    ar = gpu.Arena.from_host([arena])
    out = np.zeros(out_len)
    gpu.outer_build(ar, t, vin, out)
    ar.close()
    assert _close(out, ref)


def test_symbolic_blocking_on_device(gpu):
    """TensorFunctions::contract of the host mirror executed by BatchGEMMSeq::outer_perform == reference operators"""
    from block2_preview_amd import b2x_host
    from block2_preview_amd.planfile import read_arrays
    from test_blocking import EBLK, _sym

    for fn in EBLK:
        d = read_arrays(fn)
        _, v = b2x_host.symbolic_blocking(_sym(fn), d, True)
        assert _close(v, d["v_ref"]), fn
