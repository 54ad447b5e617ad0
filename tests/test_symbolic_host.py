"""The symbolic -> numeric host layer (block2-preview_amd/csrc/host/b2x_symbolic.hpp) against fixtures captured
from the running reference at EffectiveHamiltonian level (tests/golden/e_*.eham, oracle/ref_dump.cpp `eham=`):

* ConnectionInfo::initialize_wfn (sparse_matrix.hpp:161-289) must reproduce the reference's own arrays
  (quanta, idx, stride, ia, ib, ic bit-exact; 9j factors to 1e-14) for SZ and SU2;
* TensorFunctions::tensor_product_multiply -> OperatorFunctions::(three_)tensor_product_multiply
  (tensor_functions.hpp:1880-2025, operator_functions.hpp:474-671) must record the same number of GEMM pairs with
  the same MAC count, and the recorded plan must give the reference's sigma (checked with the CPU oracle here,
  on the device in test_symbolic_gpu.py)."""
import glob
import os

import numpy as np
import pytest

import hooks
from conftest import GOLDEN
from block2_preview_amd.planfile import PAIR_DTYPE, read_arrays
from oracle import oracle

FILES = sorted(glob.glob(os.path.join(GOLDEN, "e_*.eham")))


def sym_of(fn):
    return "su2" if "su2" in os.path.basename(fn) else "sz"


@pytest.fixture(scope="module")
def host(built):
    from block2_preview_amd import b2x_host

    return b2x_host


def test_fixtures_cover_both_symmetries_and_both_delay_sides():
    assert len(FILES) >= 5
    sides = set()
    for fn in FILES:
        d = read_arrays(fn)
        sides.add((sym_of(fn), int(d["tensor.delayed"][0]), int(d["tensor.delayed"][1])))
    assert ("su2", 1, 0) in sides and ("su2", 0, 1) in sides and ("sz", 1, 0) in sides and ("sz", 0, 1) in sides


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_initialize_wfn_matches_reference(host, fn):
    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian(sym_of(fn), d)
    c = h.wfn_cinfo()
    for k in ("n", "quanta", "idx", "stride", "ia", "ib", "ic"):
        assert np.array_equal(np.asarray(c[k]), d["wfn_cinfo." + k]), k
    assert np.abs(np.asarray(c["factor"]) - d["wfn_cinfo.factor"]).max() < 1e-14


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_recorded_plan_matches_reference(host, fn):
    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian(sym_of(fn), d)
    h.record()
    assert h.n_pairs == int(d["n_pairs"][0])  # same number of rotate / three_rotate records
    assert h.nflop == int(d["n_pairs"][1])  # same MAC count as the reference's batch[0/1]->nflop
    pairs = np.frombuffer(h.pairs().tobytes(), PAIR_DTYPE)
    sig = np.zeros(len(d["sigma_ref"]))
    oracle.replay(pairs, d["arena"], d["psi"], sig)
    assert np.abs(sig - d["sigma_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["sigma_ref"]).max())


def test_label_algebra(host):
    """SU2 / SZ label arithmetic restated field by field (symmetry.hpp:654-731, 1183-1306)"""
    mk = host.su2_make
    a, b = mk(3, 1, 1, 2), mk(4, 2, 2, 3)
    s = host.su2_add(a, b)
    assert s == mk(7, 1, 3, 1)  # spins |1-2| .. 1+2, pg xor
    dq = mk(1, 1, 1, 1)
    bra, ket = mk(5, 2, 2, 3), mk(4, 1, 1, 2)
    comb = host.su2_combine(dq, bra, ket)
    assert comb == mk(4, 2, 1, 2)  # ket with twos_low = bra spin
    assert host.su2_get_bra(comb, dq) == bra
    assert host.su2_combine(dq, mk(5, 4, 4, 3), ket) == 0xFFFFFFFFFFFFFFFF  # triangle rule violated
    z = host.sz_make(-2, -1, 5)
    assert host.sz_neg(z) == host.sz_make(2, 1, 5)
    assert host.sz_add(z, host.sz_make(3, 2, 6)) == host.sz_make(1, 1, 3)


def test_wigner_symbols(host):
    """known values: {1 1 1; 1 1 1} = 1/6, {1/2 1/2 1; 1/2 1/2 1} = 1/6; 9j with a zero reduces to a 6j"""
    assert abs(host.wigner_6j(2, 2, 2, 2, 2, 2) - 1.0 / 6.0) < 1e-14
    assert abs(host.wigner_6j(1, 1, 2, 1, 1, 2) - 1.0 / 6.0) < 1e-14
    # {a b c; d e f; g h 0} = delta(c,f) delta(g,h) (-1)^(b+c+d+g) / sqrt((2c+1)(2g+1)) {a b c; e d g}
    a, b, c, dd, e, g = 2, 2, 2, 2, 2, 2
    lhs = host.wigner_9j(a, b, c, dd, e, c, g, g, 0)
    rhs = (-1) ** ((b + c + dd + g) // 2) / np.sqrt((c + 1) * (g + 1)) * host.wigner_6j(a, b, c, e, dd, g)
    assert abs(lhs - rhs) < 1e-14


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_diagonal_terms_match_reference_diag(host, fn):
    """§8f row 1: ConnectionInfo::initialize_diag (sparse_matrix.hpp:80-159) + (three_)tensor_product_diagonal
    (operator_functions.hpp:211-328) record rank-1 terms whose sum is the reference's diag (evaluated here with numpy,
    on the device in test_symbolic_gpu.py)."""
    from block2_preview_amd.planfile import DIAG_TERM_DTYPE

    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian(sym_of(fn), d)
    t = np.frombuffer(h.diag_terms().tobytes(), DIAG_TERM_DTYPE)
    assert len(t) > 0
    diag, ar = np.zeros(len(d["diag"])), d["arena"]
    for x in t:
        a = ar[int(x["a_off"]):int(x["a_off"]) + (x["m"] - 1) * x["a_stride"] + 1:x["a_stride"]]
        b = ar[int(x["b_off"]):int(x["b_off"]) + (x["n"] - 1) * x["b_stride"] + 1:x["b_stride"]]
        idx = int(x["c_off"]) + np.arange(x["m"])[:, None] * x["ldc"] + np.arange(x["n"])[None, :]
        diag[idx] += x["alpha"] * np.outer(a, b)
    assert np.abs(diag - d["diag"]).max() <= 1e-12 * max(1.0, np.abs(d["diag"]).max())


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_diag_terms_segment_without_device(host, fn):
    """the device-side grouping of diagonal terms by psi sector accepts every fixture (incl. one-column sectors)"""
    import ctypes as C

    from block2_preview_amd import capi
    from block2_preview_amd.planfile import DIAG_TERM_DTYPE

    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian(sym_of(fn), d)
    t = np.frombuffer(h.diag_terms().tobytes(), DIAG_TERM_DTYPE).copy()
    n = C.c_uint64()
    hooks.check(hooks.lib().b2x_debug_compile_diag(C.c_size_t(len(t)), t.ctypes.data_as(C.c_void_p),
                                                 C.c_size_t(len(d["diag"])), C.c_uint64(len(d["arena"])), C.byref(n)))
    k = int(d["ket.info"][0])
    assert n.value <= len(d["info.%d.quanta" % k])  # at most one component per psi sector
