"""Blocking of the environments (blocking half): c = a (x) b element-wise block products.
Fixtures blk_*.blk: the term list re-grouped from the k = 1 GEMM groups the REFERENCE's own TensorFunctions::tensor_product
recorded (SeqTypes::Auto) for left_contract / right_contract, the block and site operators, and the enlarged operators
the reference computed (oracle/ref_dump.cpp capture_blocking).  No GPU."""
import glob
import os

import numpy as np
import pytest

import hooks
from conftest import GOLDEN
from block2_preview_amd.planfile import OUTER_TERM_DTYPE, read_arrays

BLK = sorted(glob.glob(os.path.join(GOLDEN, "blk_*.blk")))


def load_blk(fn):
    d = read_arrays(fn)
    return np.frombuffer(d["terms"].tobytes(), OUTER_TERM_DTYPE).copy(), d


def random_outer_terms(rng, n, n_sectors=5, max_dim=30):
    """windows on a grid of sub-blocks per sector (as stride selects them), plus whole-sector sums; all operand kinds"""
    in_len = arena_len = 4 * (3 * max_dim) ** 2 + 100
    sectors, off = [], 0
    for _ in range(n_sectors):
        rcuts = np.concatenate([[0], np.cumsum(rng.integers(1, max_dim, int(rng.integers(1, 4))))])
        ccuts = np.concatenate([[0], np.cumsum(rng.integers(1, max_dim, int(rng.integers(1, 4))))])
        sectors.append((off, rcuts, ccuts))
        off += int(rcuts[-1]) * int(ccuts[-1])
    out_len = off
    t = np.zeros(n, OUTER_TERM_DTYPE)
    for i in range(n):
        off, rcuts, ccuts = sectors[int(rng.integers(n_sectors))]
        ld = int(ccuts[-1])
        if rng.random() < 0.8:
            a, b = int(rng.integers(len(rcuts) - 1)), int(rng.integers(len(ccuts) - 1))
            r0, m, c0, nn = int(rcuts[a]), int(rcuts[a + 1] - rcuts[a]), int(ccuts[b]), int(ccuts[b + 1] - ccuts[b])
        else:
            r0, m, c0, nn = 0, int(rcuts[-1]), 0, ld
        kind = int(rng.integers(4))
        a_src, b_src = int(rng.integers(2)), int(rng.integers(3))
        if kind == 0:  # block * scalar
            a_rs, a_cs, b_rs, b_cs = nn + int(rng.integers(3)), 1, 0, 0
        elif kind == 1:  # transposed block * scalar
            a_rs, a_cs, b_rs, b_cs = 1, m + int(rng.integers(3)), 0, 0
        elif kind == 2:  # rank-1 (diagonal-like)
            a_rs, a_cs, b_rs, b_cs = 2, 0, 0, 3
        else:  # element-wise product of two blocks
            a_rs, a_cs, b_rs, b_cs = nn, 1, 1, m
        ea = (m - 1) * a_rs + (nn - 1) * a_cs + 1
        eb = (m - 1) * b_rs + (nn - 1) * b_cs + 1
        t[i] = (m, nn, a_rs, a_cs, b_rs, b_cs, ld, a_src, b_src, (0, 0), rng.standard_normal(),
                int(rng.integers((in_len if a_src else arena_len) - ea)),
                0 if b_src == 2 else int(rng.integers((in_len if b_src else arena_len) - eb)), off + r0 * ld + c0)
    return t, in_len, out_len, arena_len


def numpy_outer(t, arena, vin, out):
    for r in t:
        m, n = int(r["m"]), int(r["n"])

        def operand(src, off, rs, cs):
            if src == 2:
                return np.ones((m, n))
            buf = vin if src else arena
            idx = int(off) + np.arange(m)[:, None] * int(rs) + np.arange(n)[None, :] * int(cs)
            return buf[idx]

        a = operand(r["a_src"], r["a_off"], r["a_rs"], r["a_cs"])
        b = operand(r["b_src"], r["b_off"], r["b_rs"], r["b_cs"])
        c = np.lib.stride_tricks.as_strided(out[int(r["c_off"]):], (m, n), (8 * int(r["ldc"]), 8))
        c += r["alpha"] * a * b


def test_fixtures_present():
    assert len(BLK) >= 3


@pytest.mark.parametrize("seed", range(3))
def test_oracle_outer_vs_numpy(built, seed):
    from oracle import oracle

    rng = np.random.default_rng(seed)
    t, in_len, out_len, arena_len = random_outer_terms(rng, 300)
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    ref = rng.standard_normal(out_len)
    out = ref.copy()
    numpy_outer(t, arena, vin, ref)
    oracle.outer(t, arena, vin, out)
    assert np.allclose(out, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("fn", BLK, ids=os.path.basename)
def test_oracle_matches_reference_blocking(built, fn):
    """the oracle replays the terms recorded by the reference and reproduces the reference's enlarged operators"""
    from oracle import oracle

    t, d = load_blk(fn)
    out = np.zeros(int(d["lens"][3]))
    oracle.outer(t, d["arena"], d["in"], out)
    assert np.abs(out - d["out_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["out_ref"]).max())


@pytest.mark.parametrize("fn", BLK, ids=os.path.basename)
def test_compiled_outer_matches_reference(built, fn):
    """cell / work-unit compiler evaluated with host loops == reference result"""
    from block2_preview_amd import capi

    t, d = load_blk(fn)
    out = np.zeros(int(d["lens"][3]))
    nw, ne = hooks.debug_compile_and_emulate_outer(t, d["arena"], d["in"], out)
    assert nw > 0 and ne >= len(t)
    assert np.abs(out - d["out_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["out_ref"]).max())


@pytest.mark.parametrize("seed", range(4))
def test_compiled_outer_random(built, seed):
    """overlapping windows of different extents are cut into cells; every operand kind and source"""
    from block2_preview_amd import capi

    rng = np.random.default_rng(20 + seed)
    t, in_len, out_len, arena_len = random_outer_terms(rng, 400, max_dim=[8, 30, 90, 200][seed])
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    ref = rng.standard_normal(out_len)
    out = ref.copy()
    numpy_outer(t, arena, vin, ref)
    hooks.debug_compile_and_emulate_outer(t, arena, vin, out)
    assert np.allclose(out, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))


def test_outer_validation(built):
    from block2_preview_amd import capi

    t = np.zeros(1, OUTER_TERM_DTYPE)
    t[0] = (3, 4, 4, 1, 0, 0, 4, 0, 2, (0, 0), 1.0, 0, 0, 0)
    arena, vin, out = np.zeros(12), np.zeros(1), np.zeros(12)
    hooks.debug_compile_and_emulate_outer(t, arena, vin, out)
    for field, val in (("ldc", 3), ("c_off", 1), ("a_off", 1), ("a_src", 3), ("m", 0)):
        bad = t.copy()
        bad[field] = val
        with pytest.raises(capi.B2XError):
            hooks.debug_compile_and_emulate_outer(bad, arena, vin, out)


def test_host_mirror_records_block_products(built):
    """BatchGEMMSeq::tensor_product / iadd record the terms of AdvancedGEMM::tensor_product (batch_gemm.hpp:431-505)"""
    from block2_preview_amd import b2x_host

    rng = np.random.default_rng(9)
    seq = b2x_host.BatchGEMMSeq()
    out = np.zeros(1000)
    a, s = rng.random((4, 6)), np.array([[2.0]])
    seq.tensor_product(a, False, s, False, (out, 0, 10, 12), 0.5, 3 * 12 + 2)  # block * scalar into the window at (3, 2)
    seq.tensor_product(s, False, a, True, (out, 0, 10, 12), 1.5, 0)  # scalar * block^T
    assert seq.outer_dims() == [(4, 6, 6, 1, 0, 0, 12, 0.5), (6, 4, 1, 6, 0, 0, 12, 1.5)]
    b = rng.random((2, 3))
    seq.tensor_product(a, False, b, False, (out, 200, 8, 18), 1.0, 0)  # general Kronecker product: 4*6 terms of 2x3
    assert seq.n_outer == 2 + 24 and seq.outer_dims()[2] == (2, 3, 3, 1, 0, 0, 18, 1.0)
    seq.iadd((out, 500, 4, 6), a, 0.25)
    seq.iadd((out, 600, 6, 4), a, 0.25, True)
    assert seq.outer_dims()[-2:] == [(4, 6, 6, 1, 0, 0, 6, 0.25), (6, 4, 1, 6, 0, 0, 4, 0.25)]
    with pytest.raises(RuntimeError):
        seq.iadd((out, 500, 4, 6), a, 1.0, False, 0.0)
    seq.clear()
    assert seq.n_outer == 0


EBLK = sorted(glob.glob(os.path.join(GOLDEN, "blk_*.eblk")))


def _sym(fn):
    return "su2" if "su2" in os.path.basename(fn) else "sz"


@pytest.mark.parametrize("fn", EBLK, ids=os.path.basename)
def test_symbolic_blocking_oracle_result(built, fn):
    """TensorFunctions::contract -> OperatorFunctions::tensor_product of the host mirror (expressions and tensor-product
    connection infos from the reference) records block products whose oracle replay gives the reference's enlarged
    operators; term count == the re-grouped reference list when every Kronecker factor is scalar"""
    from block2_preview_amd import b2x_host
    from oracle import oracle

    d = read_arrays(fn)
    terms_b, _ = b2x_host.symbolic_blocking(_sym(fn), d, False)
    t = np.frombuffer(bytes(terms_b), OUTER_TERM_DTYPE)
    assert len(t) > 0
    v = np.zeros(int(d["meta"][7]))
    oracle.outer(t, d["site"], d["x"], v)
    assert np.abs(v - d["v_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["v_ref"]).max())
    ref_t, _ = load_blk(fn.replace(".eblk", ".blk"))
    # same number of element-term products as the list recorded by the reference
    assert int((t["m"].astype(np.int64) * t["n"]).sum()) == int((ref_t["m"].astype(np.int64) * ref_t["n"]).sum())
