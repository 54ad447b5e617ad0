"""C++ host mirror (b2x_host pybind module): recording semantics and the small dense pieces that need no GPU."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def host(built):
    from block2_preview_amd import b2x_host

    return b2x_host


def test_seqtypes_and_recording(host):
    assert int(host.SeqTypes.Tasked) == 4 and int(host.SeqTypes.Auto) == 2  # values of threading.hpp:105-135
    seq = host.BatchGEMMSeq()
    rng = np.random.default_rng(0)
    l, r = rng.random((7, 5)), rng.random((4, 9))
    # c(7x9) += 0.5 * l(7x5) a(5x4) r(4x9): conj_bra = 0, conj_ket = 2 (plain) as tensor_product_multiply passes it
    seq.rotate((0, 5, 4), (0, 7, 9), l, 0, r, 2, 0.5)
    assert seq.n_pairs == 1
    assert seq.nflop == 5 * 9 * 4 + 7 * 9 * 5  # MACs of both stages (batch_gemm.hpp:307)
    assert seq.max_work == 5 * 9
    seq.clear()
    assert seq.n_pairs == 0 and seq.nflop == 0


def test_three_rotate_needs_scalar_factor(host):
    seq = host.BatchGEMMSeq()
    x = np.ones((2, 2))
    with pytest.raises(RuntimeError):
        seq.three_rotate((0, 2, 2), (0, 2, 2), x, False, x, False, x, False, x, False, True, 1.0, 0)


@pytest.mark.parametrize("m", [1, 2, 7, 30, 63])
def test_small_eigs_matches_lapack(host, m):
    """subspace eigensolver of davidson: ascending eigenvalues, row j = eigenvector j (alpha(j, i))"""
    rng = np.random.default_rng(m)
    a = rng.standard_normal((m, m))
    a = a + a.T
    low = np.tril(a)  # davidson fills the lower triangle only
    w, v = host.small_eigs(low.ravel().tolist(), m)
    w, v = np.array(w), np.array(v).reshape(m, m)
    w_ref = np.linalg.eigvalsh(a)
    assert np.allclose(w, w_ref, atol=1e-11)
    for j in range(m):
        assert np.allclose(a @ v[j], w[j] * v[j], atol=1e-9)


def test_small_eigs_degenerate_and_diagonal(host):
    """already diagonal matrices (the state after a Davidson deflation) and repeated eigenvalues"""
    d = np.diag([3.0, -1.0, 2.0, 2.0, 2.0, 0.0])
    w, v = host.small_eigs(d.ravel().tolist(), 6)
    assert np.allclose(w, [-1, 0, 2, 2, 2, 3]) and np.allclose(np.array(v).reshape(6, 6) @ np.array(v).reshape(6, 6).T, np.eye(6))
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    a = q @ np.diag([1.0] * 5 + [2.0] * 4 + [-3.0, 7.0, 7.0]) @ q.T
    w, v = host.small_eigs(np.tril(a).ravel().tolist(), 12)
    v = np.array(v).reshape(12, 12)
    assert np.allclose(w, sorted([1.0] * 5 + [2.0] * 4 + [-3.0, 7.0, 7.0]), atol=1e-12)
    assert np.allclose(v @ v.T, np.eye(12), atol=1e-12) and np.allclose(v @ a @ v.T, np.diag(w), atol=1e-11)
