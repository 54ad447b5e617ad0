"""Parity of the HIP single-GEMM-list path (perturbative noise) through the C ABI with the oracle and
with the perturbed wavefunctions the real reference returned.  fp64, tolerance 1e-12 relative to max|result|."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from block2_preview_amd import synth
from block2_preview_amd.planfile import read_gemm_list
from oracle import oracle
from test_gemm_list import numpy_gemm_list, pnoise_files, random_gemm_list, shared_operator_list

pytestmark = pytest.mark.gpu
TOL = 1e-12
FILES = pnoise_files()
STRUCTS = sorted(glob.glob(os.path.join(GOLDEN, "*.pnoise_struct.npz")))


def _run(capi, gl, arena, vin, scale=1.0, out0=None, **kw):
    ar = capi.Arena.from_host([arena])
    plan = capi.GemmPlan(ar, gl.gemms, gl.in_len, gl.out_len, **kw)
    out = np.zeros(gl.out_len) if out0 is None else out0.copy()
    plan.execute_host(vin, out, scale)
    st = plan.stats
    plan.close(), ar.close()
    return out, st


def _close(a, b, tol=TOL):
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_golden_reference_noise(gpu, fn):
    """perturbed wavefunctions from the MFMA path == what the reference's perturbative_noise returned"""
    gl = read_gemm_list(fn)
    out, st = _run(gpu, gl, gl.arena, gl.vin)
    assert st["macs"] == gl.macs and st["macs_executed"] <= gl.macs
    assert _close(out, gl.out_ref)
    out, st = _run(gpu, gl, gl.arena, gl.vin, keep_order=1)  # record by record
    assert st["macs_executed"] == gl.macs
    assert _close(out, gl.out_ref)


@pytest.mark.parametrize("seed", range(6))
def test_random_lists_vs_oracle(gpu, seed):
    rng = np.random.default_rng(100 + seed)
    g, in_len, out_len, arena_len = random_gemm_list(rng, 500, max_dim=[20, 60, 150][seed % 3])
    gl = type("GL", (), dict(gemms=g, in_len=in_len, out_len=out_len))
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    out0 = rng.standard_normal(out_len)
    ref = out0.copy()
    oracle.gemm_list(g, arena, vin, ref, -0.75, 4)
    out, _ = _run(gpu, gl, arena, vin, -0.75, out0, item_macs=[0, 30000][seed % 2])
    assert _close(out, ref)


def test_repeatable_bitwise(gpu):
    """fixed summation order: two executions give identical bits"""
    gl = read_gemm_list(FILES[0])
    a, _ = _run(gpu, gl, gl.arena, gl.vin, item_macs=5000)
    b, _ = _run(gpu, gl, gl.arena, gl.vin, item_macs=5000)
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("fn,f", [(fn, f) for fn in STRUCTS[:2] for f in (1, 4)],
                         ids=lambda x: os.path.basename(x) if isinstance(x, str) else "x%d" % x)
def test_cr2_structure_scaled(gpu, fn, f):
    """Cr2/SVP M=250 noise list (structure captured from the reference), dimensions x f, synthetic data: vs oracle"""
    gl = synth.scale_gemm_list(read_gemm_list(fn), f)
    rng = np.random.default_rng(5)
    arena, vin = rng.random(gl.arena_len) - 0.5, rng.random(gl.in_len) - 0.5
    ref = np.zeros(gl.out_len)
    oracle.gemm_list(gl.gemms, arena, vin, ref, 1.0, 8)
    out, st = _run(gpu, gl, arena, vin)
    assert st["macs"] == gl.macs
    assert _close(out, ref)


@pytest.mark.parametrize("tb,proportional", [(0, True), (1, True), (1, False)])
def test_shared_operator_sums(gpu, tb, proportional):
    """operator sums shared between psi blocks (proportional coefficient vectors) or one per block: vs numpy, and vs the
    record-by-record replay of the same list"""
    from types import SimpleNamespace

    rng = np.random.default_rng(21 + tb)
    g, in_len, out_len, arena_len = shared_operator_list(rng, tb, proportional, k=150, n=210, n_ops=9, ms=(130, 70, 16))
    gl = SimpleNamespace(gemms=g, in_len=in_len, out_len=out_len)
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    ref = np.zeros(out_len)
    numpy_gemm_list(g, arena, vin, ref)
    out, st = _run(gpu, gl, arena, vin)
    assert st["macs_executed"] * 9 == st["macs"]
    assert _close(out, ref)
    out1, st1 = _run(gpu, gl, arena, vin, keep_order=1)
    assert st1["macs_executed"] == st1["macs"]
    assert _close(out1, ref)


def test_linearity_large(gpu):
    """size-independent property at a size the oracle does not visit: P(a x + b y) == a P(x) + b P(y)"""
    if not STRUCTS:
        pytest.skip("no Cr2 noise structure fixture")
    gl = synth.scale_gemm_list(read_gemm_list(STRUCTS[0]), 8)
    rng = np.random.default_rng(6)
    arena = rng.random(gl.arena_len) - 0.5
    x, y = rng.random(gl.in_len) - 0.5, rng.random(gl.in_len) - 0.5
    ar = gpu.Arena.from_host([arena])
    plan = gpu.GemmPlan(ar, gl.gemms, gl.in_len, gl.out_len)
    px, py, pz = np.zeros(gl.out_len), np.zeros(gl.out_len), np.zeros(gl.out_len)
    plan.execute_host(x, px), plan.execute_host(y, py), plan.execute_host(0.3 * x - 1.7 * y, pz)
    plan.close(), ar.close()
    assert _close(pz, 0.3 * px - 1.7 * py, 1e-11)


def test_host_mirror_auto_perform(gpu):
    """C++ host mirror: load the reference's recorded list, BatchGEMMSeq::auto_perform(v) == reference result; with and
    without naming the wavefunction (operands then travel as arena ranges, like every absolute pointer)"""
    from block2_preview_amd import b2x_host

    gl = read_gemm_list(FILES[-1])
    for named in (True, False):
        seq = b2x_host.BatchGEMMSeq()
        out = np.zeros(gl.out_len)
        seq.load_gemms(gl.gemms, gl.arena, gl.vin, out)
        assert seq.n_gemms == len(gl.gemms)
        seq.auto_perform(out, gl.vin if named else None)
        assert seq.n_gemms == 0 and seq.cumulative_nflop == gl.macs
        assert _close(out, gl.out_ref)


def _kron_slice(da, db):
    return np.kron(da, db)


@pytest.mark.parametrize("seed", range(4))
def test_three_rotate_tr_vs_kron(gpu, seed):
    """three_rotate_tr_left / _right vs the explicit (da x db) product, in the manner of the reference's three_rotate
    test (unit_test/test_matrix.cpp:394-462): the delayed operator acts on a row (dleft) or column (!dleft) slice"""
    from block2_preview_amd import b2x_host

    rng = np.random.default_rng(40 + seed)
    nl, nr = 3, 4  # blocks of the delayed operator along its fused index: sizes below
    dims = [int(x) for x in rng.integers(1, 6, nl)]
    big_m = sum(dims)
    n_cols = int(rng.integers(2, 9))
    other = rng.random((n_cols, n_cols))  # the non-delayed operator on the opposite side
    psi = rng.random(big_m * n_cols)
    for dleft in (True, False):
        for tr_right in (True, False):
            i, j = int(rng.integers(nl)), int(rng.integers(nl))
            if dleft != tr_right:  # the delayed operator is the traced (identity) side: same slice in and out
                j = i
            a_scalar = bool(rng.integers(2))
            blk = rng.random((dims[i], dims[j]))  # the non-scalar factor maps slice j -> slice i
            sc = np.array([[rng.standard_normal()]])
            da, db = (sc, blk) if a_scalar else (blk, sc)
            r0, c0 = sum(dims[:i]), sum(dims[:j])
            seq = b2x_host.BatchGEMMSeq()
            if dleft:
                a = (psi, 0, big_m, n_cols)
                out = np.zeros(big_m * n_cols)
                c = (out, 0, big_m, n_cols)
                bra = np.zeros((big_m, big_m))
                stride = r0 * big_m + c0  # (row offset of c, row offset of a) in the enlarged bra
                fn = seq.three_rotate_tr_right if tr_right else seq.three_rotate_tr_left
                fn(a, c, bra, False, other, False, da, False, db, False, True, 0.5, stride)
                seq.auto_perform(out, psi)
                A = psi.reshape(big_m, n_cols)
                ref = np.zeros((big_m, n_cols))
                if tr_right:  # c[r0:] += 0.5 * sc * blk . a[c0:]
                    ref[r0:r0 + dims[i]] += 0.5 * sc[0, 0] * blk @ A[c0:c0 + dims[j]]
                    assert _close(out.reshape(big_m, n_cols), ref)
                else:  # bra side traced: the row slice passes through, ket applied
                    ast, cst = stride % big_m, stride // big_m
                    h = dims[j]
                    ref[cst:cst + h] += 0.5 * A[ast:ast + h] @ other
                    assert _close(out.reshape(big_m, n_cols)[cst:cst + h], ref[cst:cst + h])
            else:
                a = (psi, 0, n_cols, big_m)
                out = np.zeros(n_cols * big_m)
                c = (out, 0, n_cols, big_m)
                ket = np.zeros((big_m, big_m))
                stride = c0 * big_m + r0  # !dleft, no conj: ast = stride / ket.n, cst = stride % ket.n
                A = psi.reshape(n_cols, big_m)
                ref = np.zeros((n_cols, big_m))
                if tr_right:  # ket side traced: c[:, cst:] += 0.5 * bra . a[:, ast:]
                    seq.three_rotate_tr_right(a, c, other, False, ket, False, da, False, db, False, False, 0.5, stride)
                    seq.auto_perform(out, psi)
                    w = dims[j]
                    ref[:, r0:r0 + w] += 0.5 * other @ A[:, c0:c0 + w]
                    assert _close(out.reshape(n_cols, big_m)[:, r0:r0 + w], ref[:, r0:r0 + w])
                else:  # c[:, cst:] += 0.5 * sc * a[:, ast:] . blk  (conj flag 2 = plain)
                    blk2 = rng.random((dims[j], dims[i]))
                    da2, db2 = (sc, blk2) if a_scalar else (blk2, sc)
                    seq.three_rotate_tr_left(a, c, other, False, ket, False, da2, False, db2, False, False, 0.5, stride)
                    seq.auto_perform(out, psi)
                    ref[:, r0:r0 + dims[i]] += 0.5 * sc[0, 0] * A[:, c0:c0 + dims[j]] @ blk2
                    assert _close(out.reshape(n_cols, big_m), ref)


def test_symbolic_perturbative_noise_on_device(gpu):
    """the symbolic walk of the host mirror, executed by BatchGEMMSeq::auto_perform == reference perturbed kets"""
    from block2_preview_amd import b2x_host
    from block2_preview_amd.planfile import read_arrays
    from test_gemm_list import ENOISE

    assert ENOISE
    for fn in ENOISE:
        d = read_arrays(fn)
        h = b2x_host.SymbolicEffectiveHamiltonian("su2" if "su2" in os.path.basename(fn) else "sz", d)
        _, v = h.perturbative_noise(d, True)
        assert _close(v, d["out_ref"]), fn


@pytest.mark.parametrize("wide", [0, 1], ids=["64-column tiles", "128-column tiles"])
def test_edge_widths_and_depths(gpu, wide):
    """Widths around the 16 / 32 / 64-column boundaries of a wave's share of a tile (a share of <= 16 columns runs the body with
    one active column fragment) and depths around the 8 / 16 boundaries of a chunk (a tail of <= 8 runs half a chunk), all four
    transposition cases, record by record against the oracle; `wide` adds one large record so that the plan takes 128-column
    tiles (4-wave workgroups) instead of 64-column ones."""
    from block2_preview_amd.planfile import GEMM_DTYPE

    rng = np.random.default_rng(7 + wide)
    dims = [(m, n, k) for m in (3, 16, 37) for n in (1, 8, 16, 17, 31, 32, 33, 48, 49, 64, 65, 80, 81, 97, 113)
            for k in (1, 7, 8, 9, 16, 17, 24, 25, 40, 41)]
    if wide:
        dims.append((130, 700, 300))
    recs, in_len, arena_len, out_len = [], 0, 0, 0
    for i, (m, n, k) in enumerate(dims):
        ta, tb = (i >> 0) & 1, (i >> 1) & 1
        lda, ldb = (m if ta else k) + (i % 3), (k if tb else n) + (i % 2)
        ea = (k - 1) * lda + m if ta else (m - 1) * lda + k
        eb = (n - 1) * ldb + k if tb else (k - 1) * ldb + n
        a_src = (i >> 2) & 1  # the other operand comes from the other buffer
        a_off = in_len if a_src else arena_len
        b_off = arena_len if a_src else in_len
        if a_src:
            in_len, arena_len = in_len + ea, arena_len + eb
        else:
            arena_len, in_len = arena_len + ea, in_len + eb
        recs.append((m, n, k, lda, ldb, n, ta, tb, a_src, 1 - a_src, 0, rng.standard_normal(), a_off, b_off, out_len))
        out_len += m * n
    g = np.zeros(len(recs), GEMM_DTYPE)
    for i, r in enumerate(recs):
        g[i] = r
    gl = type("GL", (), dict(gemms=g, in_len=in_len, out_len=out_len))
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    ref = np.zeros(out_len)
    oracle.gemm_list(g, arena, vin, ref, 1.25, 4)
    for kw in (dict(keep_order=1), dict(), dict(item_macs=20000)):
        out, st = _run(gpu, gl, arena, vin, 1.25, **kw)
        assert st["fallback"] == 0 and st["macs_issued"] >= st["macs_executed"]
        assert _close(out, ref), kw
