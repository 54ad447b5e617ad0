"""CPU checks of the host pieces of the sweep loop that need no device: the recoupling coefficients and the fused-index
bookkeeping the carried wavefunction is regrouped with (block2-preview_amd/sweep.py: _recoupling, _fuse, _connection;
reference: SparseMatrix::swap_to_fused_left / _right, src/core/sparse_matrix.hpp:1789-1927, StateInfo::get_connection_info,
src/core/state_info.hpp:283-311)."""
import numpy as np
import pytest


def _engine(sym):
    from block2_preview_amd import b2x_host
    from block2_preview_amd.sweep import DMRG

    dm = DMRG.__new__(DMRG)  # (no fixture, no device: only the label arithmetic is exercised)
    dm.sym, dm.host = sym, b2x_host
    return dm


def test_recoupling_matrix_is_orthogonal():
    """for fixed spins a, b, c (site / bond / bond) and total d the coefficients racah(a, b, c, d; e, f) sqrt((2e+1)(2f+1)) over
    the intermediate spins e of (a, b) and f of (b, c) form an orthogonal matrix: regrouping a wavefunction keeps its norm"""
    dm = _engine("su2")
    for ta, tb, tc, td in [(3, 1, 0, 4), (2, 1, 2, 3), (4, 1, 1, 4), (1, 1, 1, 1), (5, 1, 3, 3), (2, 0, 2, 0)]:
        es = [e for e in range(abs(ta - tb), ta + tb + 1, 2) if abs(e - tc) <= td <= e + tc]
        fs = [f for f in range(abs(tb - tc), tb + tc + 1, 2) if abs(ta - f) <= td <= ta + f]
        if not es or not fs:
            continue
        # swap_to_fused_left's argument order: (l, m, target, r, lm, mr) with l = a, m = b, r = c, total = d
        u = np.array([[dm._recoupling(ta, tb, td, tc, e, f) for f in fs] for e in es])
        assert len(es) == len(fs)
        assert np.abs(u @ u.T - np.eye(len(es))).max() < 1e-12, (ta, tb, tc, td)
    assert _engine("sz")._recoupling(1, 2, 3, 4, 5, 6) == 1.0


def test_fuse_and_connection_follow_block2_order():
    su2, sz = _engine("su2"), _engine("sz")
    assert su2._fuse((3, 2, 1), (1, 1, 2)) == [(4, 1, 3), (4, 3, 3)]
    assert sz._fuse((3, -1, 1), (1, 1, 2)) == [(4, 0, 3)]
    # bond (x) site, first index outermost: the pairs of one fused label keep the order (i, j) was visited in, their states
    # start where the previous pair's a_i * b_j states ended
    ak, ad = [(0, 0, 0), (1, 1, 0), (2, 0, 0)], [2, 3, 5]
    bk, bd = [(0, 0, 0), (1, 1, 0), (2, 0, 0)], [1, 1, 1]
    c = su2._connection(ak, ad, bk, bd)
    assert c[(2, 0, 0)] == [2 + 3 + 5, {(0, 2): 0, (1, 1): 2, (2, 0): 5}]
    assert c[(2, 2, 0)] == [3, {(1, 1): 0}]
    assert c[(1, 1, 0)] == [5, {(0, 1): 0, (1, 0): 2}]
    assert sorted(c) == [(0, 0, 0), (1, 1, 0), (2, 0, 0), (2, 2, 0), (3, 1, 0), (4, 0, 0)]


def test_host_cores_respects_the_container():
    from block2_preview_amd.sweep import host_cores

    assert 1 <= host_cores() <= 4096
