"""Environment rotation (rotation half): TensorFunctions::left_rotate / right_rotate as GEMM-pair
plans.  Fixtures: rot_*.plan = the pair list the REFERENCE's own tensor_rotate recorded (SeqTypes::Auto) + the rotated
operators the reference computed; rot_*.erot = the same step at the symbolic level (infos, MPS tensors).  The .plan
files are also picked up by the generic golden tests (test_oracle_golden, test_plan_compiler, test_hpsi_gpu).  No GPU."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from block2_preview_amd.planfile import PAIR_DTYPE, read_arrays, read_plan

EROT = sorted(glob.glob(os.path.join(GOLDEN, "rot_*.erot")))


def _sym(fn):
    return "su2" if "su2" in os.path.basename(fn) else "sz"


def test_fixtures_present():
    assert len(EROT) >= 3 and len(glob.glob(os.path.join(GOLDEN, "rot_*.plan"))) >= 4


@pytest.mark.parametrize("fn", EROT, ids=os.path.basename)
def test_symbolic_rotate_records_reference_pairs(built, fn):
    """tensor_rotate of the host mirror walks the sectors as the reference does: same pairs, same order"""
    from block2_preview_amd import b2x_host

    d = read_arrays(fn)
    pairs_b, _ = b2x_host.symbolic_rotate(_sym(fn), d, False)
    mine = np.frombuffer(bytes(pairs_b), PAIR_DTYPE)
    ref = read_plan(fn.replace(".erot", ".plan"))
    assert len(mine) == len(ref.pairs) == int(d["meta"][3])
    for name in PAIR_DTYPE.names:
        assert np.array_equal(mine[name], ref.pairs[name]), name


@pytest.mark.parametrize("fn", EROT, ids=os.path.basename)
def test_symbolic_rotate_oracle_result(built, fn):
    """the recorded pairs, replayed by the oracle, give the rotated operators of the reference"""
    from block2_preview_amd import b2x_host
    from oracle import oracle

    d = read_arrays(fn)
    pairs_b, _ = b2x_host.symbolic_rotate(_sym(fn), d, False)
    pairs = np.frombuffer(bytes(pairs_b), PAIR_DTYPE)
    v = np.zeros(int(d["meta"][6]))
    macs = oracle.replay(pairs, d["arena"], d["x"], v, 1.0, 4)
    assert macs == int(d["meta"][4])
    assert np.abs(v - d["v_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["v_ref"]).max())


def test_rotate_is_basis_change(built):
    """property: with a square orthogonal MPS tensor block the rotation preserves the Frobenius norm of every block"""
    from block2_preview_amd import b2x_host
    from oracle import oracle

    rng = np.random.default_rng(3)
    n = 12
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a = rng.standard_normal((n, n))
    seq = b2x_host.BatchGEMMSeq()
    # left rotate: c = q^T a q  (conj_bra = 3: transpose, conj_ket = 0)
    seq.rotate((0, n, n), (0, n, n), q, 3, q, 0, 1.0)
    assert seq.n_pairs == 1 and seq.nflop == 2 * n ** 3
    ref = q.T @ a @ q
    assert abs(np.linalg.norm(ref) - np.linalg.norm(a)) < 1e-12
