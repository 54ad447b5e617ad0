"""Restatement of the reference's executor test (unit_test/test_batch_gemm.cpp:88-143, TestRotateTasked):
a plan recorded with `rotate` and replayed must equal the direct triple product, for random dims 1..100,
1..30 outputs x 1..30 inputs and random transposes; threshold 1e-10 as in the reference."""
import numpy as np

from block2_preview_amd.planfile import PAIR_DTYPE
from oracle import oracle


def _case(rng):
    ma, na = rng.integers(1, 101, 2)
    mc, nc = rng.integers(1, 101, 2)
    ncbatch, nbatch = rng.integers(1, 31, 2)
    a = rng.random((nbatch, ma, na))
    d = rng.random(ncbatch)
    l = rng.random((mc, ma))  # bra  (mc x ma)
    r = rng.random((na, nc))  # ket  (na x nc)
    conjl, conjr = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    # storage actually handed to rotate: flipped when conj (l.flip_dims() keeps the data, swaps m/n)
    arena = np.concatenate([l.ravel(), r.ravel()])
    pairs = np.zeros(ncbatch * nbatch, PAIR_DTYPE)
    k = 0
    for ic in range(ncbatch):
        for ii in range(nbatch):
            p = pairs[k]
            # stage 0: W = a * op(ket);  ket stored (na x nc) if !conj else viewed (nc x na)
            p["m0"], p["k0"], p["n0"] = ma, na, nc
            p["lda0"] = na
            p["tb0"] = 1 if conjr else 0
            p["ldb0"] = na if conjr else nc
            # stage 1: c += d * op(bra) * W
            p["m1"], p["k1"], p["n1"] = mc, ma, nc
            p["ta1"] = 1 if conjl else 0
            p["lda1"] = mc if conjl else ma
            p["ldc1"] = nc
            p["alpha0"], p["alpha1"] = 1.0, d[ic]
            p["x_off"], p["y_off"], p["z_off"], p["v_off"] = ii * ma * na, l.size, 0, ic * mc * nc
            k += 1
    L = l.reshape(ma, mc).T if conjl else l
    R = r.reshape(nc, na).T if conjr else r
    std = np.stack([d[ic] * sum(L @ a[ii] @ R for ii in range(nbatch)) for ic in range(ncbatch)])
    return pairs, arena, a.ravel(), std.ravel()


def test_rotate_tasked_random_cases():
    rng = np.random.default_rng(1969)
    for _ in range(40):
        pairs, arena, psi, std = _case(rng)
        sig = np.zeros(std.size)
        oracle.replay(pairs, arena, psi, sig, 1.0, int(rng.integers(1, 5)))
        assert np.allclose(sig, std, rtol=1e-10, atol=1e-10)
