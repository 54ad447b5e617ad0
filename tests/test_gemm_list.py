"""Single-GEMM lists (perturbative noise): oracle vs the reference's own result, plan compiler vs
oracle, host-mirror recording.  No GPU needed."""
import glob
import os

import numpy as np
import pytest

import hooks
from conftest import GOLDEN


def pnoise_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "*.pnoise")))


def random_gemm_list(rng, n, n_sectors=6, max_dim=40):
    """Random records in the shape of a noise list: output sectors (rows x cols, stacked row slices), each record
    writes a full-width row slice or a column slice; operands from the arena / input vector, both transposes."""
    from block2_preview_amd.planfile import GEMM_DTYPE

    in_len = 2 * (3 * max_dim + 3) * (max_dim + 3)
    arena_len = 3 * in_len
    sectors, off = [], 0
    for _ in range(n_sectors):
        rows, cols = int(rng.integers(1, max_dim * 3)), int(rng.integers(1, max_dim * 3))
        cuts = sorted(set([0, rows] + [int(x) for x in rng.integers(0, rows + 1, 2)]))
        sectors.append((off, rows, cols, cuts))
        off += rows * cols
    out_len = off
    g = np.zeros(n, GEMM_DTYPE)
    for i in range(n):
        off, rows, cols, cuts = sectors[int(rng.integers(n_sectors))]
        k = int(rng.integers(1, max_dim))
        if rng.random() < 0.6:  # row slice, all columns
            j = int(rng.integers(len(cuts) - 1))
            r0, m, c0, nn = cuts[j], cuts[j + 1] - cuts[j], 0, cols
        else:  # column slice, all rows
            c0 = int(rng.integers(cols))
            nn = int(rng.integers(1, cols - c0 + 1))
            r0, m = 0, rows
        ta, tb = int(rng.integers(2)), int(rng.integers(2))
        a_src, b_src = int(rng.integers(2)), int(rng.integers(2))
        lda = (m if ta else k) + int(rng.integers(3))
        ldb = (k if tb else nn) + int(rng.integers(3))
        ea = (k - 1) * lda + m if ta else (m - 1) * lda + k
        eb = (nn - 1) * ldb + k if tb else (k - 1) * ldb + nn
        g[i] = (m, nn, k, lda, ldb, cols, ta, tb, a_src, b_src, 0, rng.standard_normal(),
                int(rng.integers((in_len if a_src else arena_len) - ea)),
                int(rng.integers((in_len if b_src else arena_len) - eb)), off + r0 * cols + c0)
    return g, in_len, out_len, arena_len


def numpy_gemm_list(g, arena, vin, out, scale=1.0):
    for r in g:
        A = vin if r["a_src"] else arena
        B = vin if r["b_src"] else arena
        m, n, k = int(r["m"]), int(r["n"]), int(r["k"])
        if r["ta"]:
            a = np.lib.stride_tricks.as_strided(A[int(r["a_off"]):], (k, m), (8 * int(r["lda"]), 8)).T
        else:
            a = np.lib.stride_tricks.as_strided(A[int(r["a_off"]):], (m, k), (8 * int(r["lda"]), 8))
        if r["tb"]:
            b = np.lib.stride_tricks.as_strided(B[int(r["b_off"]):], (n, k), (8 * int(r["ldb"]), 8)).T
        else:
            b = np.lib.stride_tricks.as_strided(B[int(r["b_off"]):], (k, n), (8 * int(r["ldb"]), 8))
        c = np.lib.stride_tricks.as_strided(out[int(r["c_off"]):], (m, n), (8 * int(r["ldc"]), 8))
        c += scale * r["alpha"] * (a @ b)


def test_gemm_dtype_matches_header():
    from block2_preview_amd.planfile import GEMM_DTYPE

    assert GEMM_DTYPE.itemsize == 64  # sizeof(b2x_gemm), include/b2x.h
    assert GEMM_DTYPE.fields["alpha"][1] == 32 and GEMM_DTYPE.fields["c_off"][1] == 56


@pytest.mark.parametrize("seed,nthreads", [(0, 1), (1, 3), (2, 8)])
def test_oracle_gemm_list_vs_numpy(built, seed, nthreads):
    from oracle import oracle

    rng = np.random.default_rng(seed)
    g, in_len, out_len, arena_len = random_gemm_list(rng, 300)
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    ref = rng.standard_normal(out_len)
    out = ref.copy()
    numpy_gemm_list(g, arena, vin, ref, 0.7)
    macs = oracle.gemm_list(g, arena, vin, out, 0.7, nthreads)
    assert macs == int((g["m"].astype(np.int64) * g["n"] * g["k"]).sum())
    assert np.allclose(out, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("fn", pnoise_files(), ids=os.path.basename)
def test_oracle_matches_reference_noise(built, fn):
    """The oracle replays the list the REFERENCE recorded and reproduces the perturbed wavefunctions the reference's own
    EffectiveHamiltonian::perturbative_noise returned (oracle/ref_dump.cpp capture_pnoise)."""
    from block2_preview_amd.planfile import read_gemm_list
    from oracle import oracle

    gl = read_gemm_list(fn)
    assert gl.out_ref is not None and len(gl.gemms) > 0
    out = np.zeros(gl.out_len)
    macs = oracle.gemm_list(gl.gemms, gl.arena, gl.vin, out, 1.0, 4)
    assert macs == gl.macs
    assert np.abs(out - gl.out_ref).max() <= 1e-12 * max(1.0, np.abs(gl.out_ref).max())


@pytest.mark.parametrize("fn", pnoise_files(), ids=os.path.basename)
def test_compiled_gemm_list_matches_reference(built, fn):
    """plan compiler (tiles, segments, items, slabs) evaluated with host loops == reference result"""
    from block2_preview_amd import capi
    from block2_preview_amd.planfile import read_gemm_list

    gl = read_gemm_list(fn)
    out = np.zeros(gl.out_len)
    st = hooks.debug_compile_and_emulate_gemms(gl.gemms, gl.in_len, gl.out_len, gl.arena, gl.vin, out, keep_order=1)
    assert st["macs"] == gl.macs and st["macs_executed"] == gl.macs  # record by record, as the reference replays it
    assert np.abs(out - gl.out_ref).max() <= 1e-12 * max(1.0, np.abs(gl.out_ref).max())
    # shipped: operator blocks that meet the same psi block in the same output window are summed first
    out = np.zeros(gl.out_len)
    st = hooks.debug_compile_and_emulate_gemms(gl.gemms, gl.in_len, gl.out_len, gl.arena, gl.vin, out)
    assert st["macs"] == gl.macs and st["macs_executed"] <= gl.macs
    assert np.abs(out - gl.out_ref).max() <= 1e-12 * max(1.0, np.abs(gl.out_ref).max())


@pytest.mark.parametrize("seed,item_macs", [(3, 0), (4, 20000), (5, 1)])
def test_compiled_gemm_list_random(built, seed, item_macs):
    from block2_preview_amd import capi

    rng = np.random.default_rng(seed)
    g, in_len, out_len, arena_len = random_gemm_list(rng, 400, max_dim=120)
    arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
    ref = rng.standard_normal(out_len)
    out = ref.copy()
    numpy_gemm_list(g, arena, vin, ref, -1.3)
    st = hooks.debug_compile_and_emulate_gemms(g, in_len, out_len, arena, vin, out, -1.3, item_macs=item_macs)
    assert st["n_items"] >= st["n_tiles"] > 0
    assert np.allclose(out, ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()))


def shared_operator_list(rng, tb, proportional, k=37, n=53, n_ops=12, ms=(40, 24, 31)):
    """n_ops operator blocks (k x n, stored transposed if tb) applied to len(ms) psi blocks, one output window each; the
    coefficient vector of psi block j is (0.5 + j) x a common vector, or independent"""
    from block2_preview_amd.planfile import GEMM_DTYPE

    ldb = (k if tb else n) + 2
    op_len = (n if tb else k) * ldb
    x_off, c_off, in_len, out_len = [], [], 0, 0
    for m in ms:
        x_off.append(in_len), c_off.append(out_len)
        in_len, out_len = in_len + m * k, out_len + m * n
    base = rng.standard_normal(n_ops)
    g = np.zeros(n_ops * len(ms), GEMM_DTYPE)
    for j, m in enumerate(ms):
        al = base * (0.5 + j) if proportional else rng.standard_normal(n_ops)
        for i in range(n_ops):
            g[j * n_ops + i] = (m, n, k, k, ldb, n, 0, tb, 1, 0, 0, al[i], x_off[j], i * op_len, c_off[j])
    rng.shuffle(g)
    return g, in_len, out_len, n_ops * op_len


@pytest.mark.parametrize("tb", [0, 1])
def test_operator_sums_are_shared_between_psi_blocks(built, tb):
    """The same operator blocks applied to several psi blocks with proportional coefficient vectors (the coupling factor
    of the psi block) need ONE sum S = sum_i alpha_i B_i; independent coefficient vectors need one sum per psi block."""
    from block2_preview_amd import capi

    rng = np.random.default_rng(11 + tb)
    k, n, n_ops, ms = 37, 53, 12, (40, 24, 31)
    res = {}
    for proportional in (True, False):
        g, in_len, out_len, arena_len = shared_operator_list(rng, tb, proportional, k, n, n_ops, ms)
        arena, vin = rng.standard_normal(arena_len), rng.standard_normal(in_len)
        ref, out = np.zeros(out_len), np.zeros(out_len)
        numpy_gemm_list(g, arena, vin, ref)
        st = hooks.debug_compile_and_emulate_gemms(g, in_len, out_len, arena, vin, out)
        assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
        assert st["macs_executed"] == sum(m * n * k for m in ms)  # one product per psi block
        res[proportional] = st["device_bytes"]
    per_sum = ((k * n + 1) & ~1) * 8 + n_ops * 48  # S + its entry list
    assert res[False] - res[True] >= 2 * per_sum  # two sums fewer


def test_gemm_list_validation(built):
    from block2_preview_amd import capi
    from block2_preview_amd.planfile import GEMM_DTYPE

    g = np.zeros(1, GEMM_DTYPE)
    g[0] = (4, 4, 4, 4, 4, 4, 0, 0, 0, 1, 0, 1.0, 0, 0, 0)
    arena, vin, out = np.zeros(16), np.zeros(16), np.zeros(16)
    hooks.debug_compile_and_emulate_gemms(g, 16, 16, arena, vin, out)
    for field, val in (("lda", 3), ("c_off", 1), ("b_off", 1), ("ta", 2), ("k", 0)):
        bad = g.copy()
        bad[field] = val
        with pytest.raises(capi.B2XError):
            hooks.debug_compile_and_emulate_gemms(bad, 16, 16, arena, vin, out)


def test_struct_fixture_roundtrip(built, tmp_path):
    from block2_preview_amd.planfile import read_gemm_list, write_gemm_struct_npz

    files = pnoise_files()
    if not files:
        pytest.skip("no pnoise fixtures")
    gl = read_gemm_list(files[0])
    fn = str(tmp_path / "x.pnoise_struct.npz")
    write_gemm_struct_npz(fn, gl)
    g2 = read_gemm_list(fn)
    assert g2.gemms.tobytes() == gl.gemms.tobytes() and g2.out_len == gl.out_len and g2.macs == gl.macs


def test_host_mirror_records_noise_gemms(built):
    """BatchGEMMSeq::multiply / three_rotate_tr_left / three_rotate_tr_right record the xgemm slots of
    src/core/batch_gemm.hpp:887-891, 1025-1109 (dims, transposes, leading dimensions, alpha)."""
    from block2_preview_amd import b2x_host

    rng = np.random.default_rng(7)
    seq = b2x_host.BatchGEMMSeq()
    psi, out = rng.random(400), np.zeros(400)
    op = rng.random((5, 7))
    # TraceTypes::Right: v(5 x 6) += 0.5 * op(5x7) . c(7x6)
    seq.multiply(op, 0, (psi, 10, 7, 6), 0, (out, 0, 5, 6), 0.5, 1.0)
    # TraceTypes::Left with a transposed right operator (conj flag 1): v(4 x 5) += c(4x7) . op^T
    seq.multiply((psi, 0, 4, 7), 0, op, 1, (out, 30, 4, 5), 2.0, 1.0)
    assert seq.gemm_dims() == [(0, 0, 5, 6, 7, 7, 6, 6, 0.5), (0, 1, 4, 5, 7, 7, 7, 5, 2.0)]
    with pytest.raises(RuntimeError):
        seq.multiply(op, 0, (psi, 10, 7, 6), 0, (out, 0, 5, 6), 0.5, 0.0)  # only accumulation is recorded
    seq.clear()
    # delayed left operator da (x) db with db = 1x1: rows [stride / bra.n ...) of a sector
    da, db = rng.random((3, 3)), np.array([[1.5]])
    bra = np.zeros((6, 6))  # only its width enters (stride decoding)
    ket = rng.random((8, 8))
    seq2 = b2x_host.BatchGEMMSeq()
    seq2.three_rotate_tr_right((psi, 0, 6, 8), (out, 0, 6, 8), bra, False, ket, False, da, False, db, False, True, 2.0,
                               3 * 6 + 3)
    # ast = stride % bra.n = 3, cst = stride / bra.n = 3: c[3:6] += (2.0 * 1.5) * da . a[3:6]
    assert seq2.gemm_dims() == [(0, 0, 3, 8, 3, 3, 8, 8, 3.0)]
    seq2.three_rotate_tr_left((psi, 0, 6, 8), (out, 0, 6, 8), bra, False, ket, True, da, False, db, False, True, 1.0,
                              3 * 6 + 3)
    assert seq2.gemm_dims()[1] == (0, 1, 3, 8, 8, 8, 8, 8, 1.0)
    assert seq2.n_gemms == 2


ENOISE = sorted(glob.glob(os.path.join(GOLDEN, "*.enoise")))


@pytest.mark.parametrize("fn", ENOISE, ids=os.path.basename)
def test_symbolic_perturbative_noise(built, fn):
    """SymbolicEffectiveHamiltonian::perturbative_noise of the host mirror (initialize_wfn per (sub-label, target),
    tensor_product_partial_multiply walk, TraceTypes variants of (three_)tensor_product_multiply) records as many GEMMs
    and MACs as the reference did and its oracle replay gives the reference's perturbed wavefunctions"""
    from block2_preview_amd import b2x_host
    from block2_preview_amd.planfile import GEMM_DTYPE, read_arrays
    from oracle import oracle

    d = read_arrays(fn)
    h = b2x_host.SymbolicEffectiveHamiltonian("su2" if "su2" in os.path.basename(fn) else "sz", d)
    gb, _ = h.perturbative_noise(d, False)
    g = np.frombuffer(bytes(gb), GEMM_DTYPE)
    assert len(g) == int(d["noise.args"][3])
    out = np.zeros(int(d["noise.args"][5]))
    macs = oracle.gemm_list(g, d["arena"], d["psi"], out)
    assert macs == int(d["noise.args"][4])
    assert np.abs(out - d["out_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["out_ref"]).max())
