"""Parity of the HIP H·psi path (through the C ABI) with the CPU oracle and with the golden vectors
captured from the real reference.  fp64: tolerance 1e-12 relative to max|sigma| (the reference's own
replay-vs-direct test uses 1e-10, unit_test/test_batch_gemm.cpp:88-143)."""
import os

import numpy as np
import pytest

from conftest import fill_plan, golden_plan_files
from block2_preview_amd import synth
from block2_preview_amd.planfile import read_plan
from oracle import oracle

pytestmark = pytest.mark.gpu
FILES = golden_plan_files()
TOL = 1e-12


def _run(capi, pf, scale=1.0, sigma0=None, **kw):
    arena = capi.Arena.from_host([pf.arena])
    plan = capi.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len, **kw)
    sig = np.zeros(pf.sigma_len) if sigma0 is None else sigma0.copy()
    plan.execute_host(pf.psi, sig, scale)
    st = plan.stats
    plan.close(), arena.close()
    return sig, st


def _close(a, b):
    return np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_golden_reference_sigma(gpu, fn):
    """sigma from the MFMA path == sigma the reference itself computed for this plan"""
    pf = read_plan(fn)
    sig, st = _run(gpu, pf)
    assert st["macs"] == int(pf.meta[5])
    assert _close(sig, pf.sigma_ref)


@pytest.mark.parametrize("fn", FILES[:2], ids=[os.path.basename(f) for f in FILES[:2]])
def test_golden_generic_kernel(gpu, fn):
    """the generic (atomic) kernel is an independent on-device cross-check"""
    pf = read_plan(fn)
    sig, _ = _run(gpu, pf, kernel=1)
    assert np.abs(sig - pf.sigma_ref).max() <= 1e-11 * max(1.0, np.abs(pf.sigma_ref).max())


@pytest.mark.parametrize("seed", range(8))
def test_random_plans_vs_oracle(gpu, seed):
    """shape family of the reference's TestRotateTasked (dims 1..100, random transposes) + row/col slices"""
    rng = np.random.default_rng(1969 + seed)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=5, max_dim=100, max_terms=12), seed)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 4)
    sig, _ = _run(gpu, pf)
    assert _close(sig, ref)


@pytest.mark.parametrize("tile_n,item_macs", [(16, 1), (32, 30000), (64, 0), (128, 0), (128, 1 << 40)])
def test_every_kernel_class(gpu, tile_n, item_macs):
    """all template instances (1/2/4/8 waves) and split granularities give the oracle's answer"""
    rng = np.random.default_rng(11)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=3, max_dim=300, max_terms=4), 11)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 8)
    sig, st = _run(gpu, pf, tile_n=tile_n, item_macs=item_macs)
    assert _close(sig, ref), st


def test_accumulate_scale_linearity_determinism(gpu):
    pf = read_plan([f for f in FILES if "h10szm50.sw1.site5" in f][0])
    base, _ = _run(gpu, pf)
    # sigma += scale * H psi   (BatchGEMMSeq::operator()(c, v, scale): beta = 1 on stage 1)
    s0 = np.linspace(-1, 1, pf.sigma_len)
    acc, _ = _run(gpu, pf, scale=-0.25, sigma0=s0)
    assert _close(acc, s0 - 0.25 * base)
    # linearity in psi
    rng = np.random.default_rng(2)
    psi2 = rng.random(pf.psi_len)
    pf2 = read_plan([f for f in FILES if "h10szm50.sw1.site5" in f][0])
    pf2.psi = psi2
    s2, _ = _run(gpu, pf2)
    pf2.psi = 2.0 * pf.psi - 3.0 * psi2
    s3, _ = _run(gpu, pf2)
    assert _close(s3, 2.0 * base - 3.0 * s2)
    # no atomics: bitwise identical run to run
    again, _ = _run(gpu, pf)
    assert np.array_equal(again, base)


def test_hermitian_effective_hamiltonian(gpu):
    """H_eff of a ground-state DMRG site is symmetric: <x|H y> == <H x|y> (size-independent property)"""
    pf = read_plan([f for f in FILES if "n2su2.sw2.site5" in f][0])
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal(pf.psi_len), rng.standard_normal(pf.psi_len)
    pf.psi = x
    hx, _ = _run(gpu, pf)
    pf.psi = y
    hy, _ = _run(gpu, pf)
    # SU2 reduced wavefunctions carry no extra metric in block2's two-site basis
    assert abs(x @ hy - hx @ y) <= 1e-10 * max(1.0, abs(x @ hy))


def _struct(name):
    from block2_preview_amd.planfile import read_struct_npz

    return read_struct_npz(os.path.join(os.path.dirname(FILES[0]), name))


def _vs_oracle(gpu, pf, seed, threads=16, **kw):
    pf = fill_plan(pf, seed)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, threads)
    sig, st = _run(gpu, pf, **kw)
    assert st["fallback"] == 0 and st["macs"] == pf.macs
    assert _close(sig, ref), st
    return sig, st


def test_config1_h10_m500_true_structure(gpu):
    """BASELINE configs[1] at its TRUE size: the H.psi plan the reference records for H10/STO-6G, SZ, M=500 at the
    mid-chain site (8 276 pairs, 0.48 GMAC: tests/golden/h10_sz_m500_sw1_site4.struct.npz, captured by oracle/ref_dump
    from the running reference), synthetic operator / psi data, against the oracle on every path."""
    pf = _struct("h10_sz_m500_sw1_site4.struct.npz")
    assert len(pf.pairs) == 8276 and pf.macs == 483096896
    sig, st = _vs_oracle(gpu, pf, 9)
    _vs_oracle(gpu, pf, 9, keep_order=1)
    _vs_oracle(gpu, pf, 9, two_stage=-1)  # all sectors on the fused wave kernel
    again, _ = _run(gpu, pf)
    assert np.array_equal(again, sig)


def test_config1_scale_h10_m500(gpu):
    """the H10 structure captured WITH DATA at M=50 (golden, sigma_ref inside) scaled x10 to the M=500 block sizes,
    random data, against the oracle (2.6 GMAC replay)"""
    pf = read_plan([f for f in FILES if "h10szm50.sw0.site6" in f][0])
    _vs_oracle(gpu, synth.scale_plan(pf, 10), 9)


def test_config5_hubbard_l16_m3000_true_structure(gpu):
    """BASELINE configs[4] at its TRUE size: the plan the reference records for the 1D Hubbard chain L=16, U/t=4, SZ at
    M=3000 (sweep 0 from MPS::random_canonicalize at fixed M, two-site step 7-8: 692 pairs, 54 GMAC, 415 MB of operators,
    psi of 6.4 M elements; tests/golden/hubbard_l16_u4_sz_m3000_sw0_site7.struct.npz), synthetic data.  The oracle replays
    the FULL plan on the host cores (about 10 s), so parity at this size is direct, not by properties."""
    pf = _struct("hubbard_l16_u4_sz_m3000_sw0_site7.struct.npz")
    assert len(pf.pairs) == 692 and pf.macs == 53960716386
    pf = fill_plan(pf, 21)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 16)
    sig, st = _run(gpu, pf)
    assert st["fallback"] == 0 and st["macs_issued"] > 0
    assert _close(sig, ref), st
    sig2, st2 = _run(gpu, pf, keep_order=1)
    assert st2["macs_executed"] == pf.macs and _close(sig2, ref), st2
    again, _ = _run(gpu, pf)
    assert np.array_equal(again, sig)


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_golden_two_stage_path(gpu, fn):
    """the grouped-GEMM (two-stage, W through scratch) path on the reference's golden plans"""
    pf = read_plan(fn)
    sig, st = _run(gpu, pf, two_stage=1, keep_order=1)
    assert st["macs_issued"] > 0 and st["macs_executed"] == st["macs"]
    assert _close(sig, pf.sigma_ref)
    sig, st = _run(gpu, pf, two_stage=1)  # per pair the cheaper of (op(Z) X) op(Y) and op(Z) (X op(Y))
    assert st["macs_executed"] <= st["macs"]
    assert _close(sig, pf.sigma_ref)


@pytest.mark.parametrize("seed,scratch_mb,item_macs", [(0, 0, 0), (1, 1, 0), (2, 1, 200000), (3, 2, 1 << 40)])
def test_two_stage_random_multi_superstep(gpu, seed, scratch_mb, item_macs):
    """several super-steps (1-2 MiB of W scratch), several items per tile, row/col slices, tiles > 256 rows"""
    rng = np.random.default_rng(300 + seed)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=3, max_dim=420, max_terms=5), seed)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 8)
    sig, st = _run(gpu, pf, two_stage=1, scratch_mb=scratch_mb, item_macs=item_macs)
    assert _close(sig, ref), st
    again, _ = _run(gpu, pf, two_stage=1, scratch_mb=scratch_mb, item_macs=item_macs)
    assert np.array_equal(again, sig)


def test_auto_routing_mixed_paths(gpu):
    """auto mode: sectors taller than one fused tile go two-stage, the rest stay fused — same answer"""
    pf = read_plan([f for f in FILES if "h10szm50.sw2.site4" in f][0])
    big = fill_plan(synth.scale_plan(pf, 8), 4)
    ref = np.zeros(big.sigma_len)
    oracle.replay(big.pairs, big.arena, big.psi, ref, 1.0, 8)
    sig, st = _run(gpu, big)
    assert _close(sig, ref), st


def test_edge_cases(gpu):
    """empty plan, 1x1x1 pair, leading dimensions larger than the rows, a pair deeper than the fused path takes (k0 > 512)"""
    from block2_preview_amd.planfile import PAIR_DTYPE

    arena = gpu.Arena.from_host([np.arange(1.0, 5000.0)])
    plan = gpu.Plan(arena, np.zeros(0, PAIR_DTYPE), 7, 5)
    sig = np.full(5, 3.0)
    plan.execute_host(np.ones(7), sig, 2.0)
    assert np.array_equal(sig, np.full(5, 3.0)) and plan.stats["macs"] == 0
    plan.close()
    p = np.zeros(3, PAIR_DTYPE)
    # 1 x 1 x 1:  v[2] += 0.5 * z * (x * y)
    p[0] = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, 0.5, 3, 10, 20, 2)
    # 3x2 block of psi with lda 5, op(Y) = Y^T (2x4 stored 4x2, ldb 3), Z 2x3 with lda 4, window of sigma with ldc 6
    p[1] = (3, 4, 2, 5, 3, 2, 4, 3, 4, 6, 0, 1, 0, 0, 0, 1.5, -1.0, 10, 100, 200, 10)
    # deep pair: k0 = 600 (> 512) goes through the grouped-GEMM path
    p[2] = (5, 3, 600, 600, 3, 4, 3, 5, 5, 3, 0, 0, 0, 0, 0, 1.0, 1.0, 40, 300, 2400, 40)
    rng = np.random.default_rng(12)
    psi_len, sigma_len = 40 + 5 * 600, 60
    psi = rng.random(psi_len)
    arena_h = rng.random(5000)
    ar = gpu.Arena.from_host([arena_h])
    plan = gpu.Plan(ar, p, psi_len, sigma_len)
    sig = rng.random(sigma_len)
    ref = sig.copy()
    oracle.replay(p, arena_h, psi, ref, 0.25, 1)
    plan.execute_host(psi, sig, 0.25)
    assert _close(sig, ref)
    assert plan.stats["macs_issued"] > 0  # the deep pair took the two-stage path
    plan.close(), ar.close(), arena.close()


def test_true_cr2_m1000_structure_against_the_oracle(gpu):
    """the plan the reference records for Cr2/SVP SU2 at its TRUE M=1000 (sweep 1, site 20: 206 245 pairs, 371 GMAC, 5.2 GB of
    operators; tests/golden/cr2_su2_m1000_sw1_site20.struct.npz), synthetic data: the oracle replays the FULL plan on the
    host cores, so parity at this size is direct.  Default compilation (association, shared products, sum pass) and the
    reference's order pair by pair."""
    pf = _struct("cr2_su2_m1000_sw1_site20.struct.npz")
    assert len(pf.pairs) == 206245 and pf.macs == 371375030711
    pf = fill_plan(pf, 33)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 16)
    sig, st = _run(gpu, pf)
    assert st["fallback"] == 0 and st["macs_issued"] > 0 and st["n_shared_products"] > 0 and st["n_flipped"] > 0
    assert _close(sig, ref), st
    sig2, st2 = _run(gpu, pf, keep_order=1)
    assert st2["macs_executed"] == pf.macs and _close(sig2, ref), st2


@pytest.mark.parametrize("sfile,scale", [("cr2_su2_m250_sw1_site20.struct.npz", 8), ("cr2_su2_m250_sw1_site20.struct.npz", 16),
                                         ("cr2_su2_m2000_sw1_site20.struct.npz", 1), ("cr2_su2_m2000_sw1_site20.struct.npz", 2)],
                         ids=["cr2_m2000", "cr2_m4000", "cr2_true_m2000", "cr2_true_m4000"])
def test_full_size_cr2_properties(gpu, sfile, scale):
    """BASELINE sizes of the Cr2/SVP SU2 mid-chain plan: x8 -> M=2000 (configs[2]: 2.6 TMAC, 9.2 GB of operators) and
    x16 -> M=4000 (configs[3], the bench workload: 98 722 pairs, 20.7 TMAC, 73 GB of operators).  The oracle cannot visit
    these, so the MFMA path is checked through size-independent properties: linearity in psi, bitwise repeatability,
    agreement with the reference's order of operations (keep_order = 1) and with the independent per-pair atomic kernel
    (hpsi_generic) on the same device data."""
    import torch

    from block2_preview_amd.planfile import read_struct_npz

    base = read_struct_npz(os.path.join(os.path.dirname(FILES[0]), sfile))  # (cr2_true_*: the reference's capture at its
    full = synth.scale_plan(base, scale) if scale > 1 else base              #  TRUE M=2000; x2 = the default bench workload)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    arena_t = torch.empty(full.arena_len, dtype=torch.float64, device=dev)
    for a in range(0, full.arena_len, 1 << 28):
        arena_t[a:a + (1 << 28)].uniform_(-0.5, 0.5, generator=g)
    x = torch.empty(full.psi_len, dtype=torch.float64, device=dev).uniform_(-0.5, 0.5, generator=g)
    y = torch.empty(full.psi_len, dtype=torch.float64, device=dev).uniform_(-0.5, 0.5, generator=g)
    z = 0.3 * x - 1.7 * y
    arena = gpu.Arena.adopt_device(arena_t.data_ptr(), full.arena_len, keep=arena_t)
    plan = gpu.Plan(arena, full.pairs, full.psi_len, full.sigma_len)
    assert plan.stats["macs"] == full.macs and plan.stats["fallback"] == 0 and plan.stats["n_staged"] <= 8
    s = torch.cuda.current_stream().cuda_stream

    def apply(v, **kw):
        out = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
        plan.execute_device(v.data_ptr(), out.data_ptr(), 1.0, s)
        torch.cuda.synchronize()
        return out

    hx, hy, hz = apply(x), apply(y), apply(z)
    scale = float(hz.abs().max())
    assert float((hz - (0.3 * hx - 1.7 * hy)).abs().max()) <= 1e-11 * scale
    assert torch.equal(apply(x), hx)  # fixed summation order
    assert plan.stats["macs_executed"] < 0.7 * full.macs  # association / sharing / sums of products
    assert plan.stats["n_merged_groups"] > 0 and plan.stats["n_shared_products"] > 0 and plan.stats["n_flipped"] > 0
    plan.close()
    # the same plan replayed pair by pair in the reference's order of operations
    ref_order = gpu.Plan(arena, full.pairs, full.psi_len, full.sigma_len, keep_order=1)
    assert ref_order.stats["macs_executed"] == full.macs
    out = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
    ref_order.execute_device(x.data_ptr(), out.data_ptr(), 1.0, s)
    torch.cuda.synchronize()
    assert float((out - hx).abs().max()) <= 1e-11 * float(hx.abs().max())
    ref_order.close()
    generic = gpu.Plan(arena, full.pairs, full.psi_len, full.sigma_len, kernel=1)
    out = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
    generic.execute_device(x.data_ptr(), out.data_ptr(), 1.0, s)
    torch.cuda.synchronize()
    assert float((out - hx).abs().max()) <= 1e-10 * float(hx.abs().max())
    generic.close(), arena.close()


@pytest.mark.parametrize("seed,scratch_mb,keep_order", [(0, 0, 0), (1, 0, 1), (2, 1, 0), (3, 1, 0), (4, 2, 0)])
def test_shared_products_and_association(gpu, seed, scratch_mb, keep_order):
    """operator-product plans (shared stage-0 products, both associations, sharing groups split over super-steps) vs the
    pair-by-pair oracle; repeatable bit for bit"""
    rng = np.random.default_rng(500 + seed)
    pf = fill_plan(synth.operator_product_plan(rng, n_row=3, n_col=4, max_dim=[60, 150, 300, 260, 420][seed], n_left=4,
                                               n_right=3, n_terms=9), seed)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 8)
    sig, st = _run(gpu, pf, scratch_mb=scratch_mb, keep_order=keep_order)
    assert _close(sig, ref), st
    if not keep_order and st["macs_issued"]:
        assert st["macs_executed"] < st["macs"]
    again, _ = _run(gpu, pf, scratch_mb=scratch_mb, keep_order=keep_order)
    assert np.array_equal(again, sig)
    if not keep_order:  # operator pre-sums formed at plan creation
        pre, st3 = _run(gpu, pf, scratch_mb=scratch_mb, presum=1)
        assert _close(pre, ref), st3


@pytest.mark.parametrize("wide", [0, 1], ids=["narrow plan", "wide plan"])
def test_edge_widths_and_depths(gpu, wide):
    """Pairs whose output widths sit around the 16 / 32 / 64-column boundaries of a wave's share of a tile and whose depths
    (k0, k1 = m0) sit around the 8 / 16 boundaries of a chunk: the one-column-fragment body and the half-depth tail chunk of
    gg_kernel in every workgroup class (1-, 2- and 4-wave: `wide` adds large pairs so that the plan takes 128-column tiles),
    both transpositions of both operators, against the oracle, in the reference's order and in the compiler's."""
    from block2_preview_amd.planfile import PAIR_DTYPE

    rng = np.random.default_rng(21 + wide)
    dims = [(m1, n, k0, k1) for m1 in (5, 16, 40) for n in (1, 8, 16, 17, 32, 33, 49, 64, 65, 81, 113)
            for k0 in (1, 8, 9, 17, 24, 41) for k1 in (3, 8, 16, 25)]
    if wide:
        dims += [(150, 600, 200, 140)] * 2
    recs, psi_len, arena_len, sigma_len = [], 0, 0, 0
    for i, (m1, n, k0, k1) in enumerate(dims):
        tb0, ta1 = i & 1, (i >> 1) & 1
        lda0 = k0 + (i % 3)                      # X: k1 x k0
        ldb0 = (k0 if tb0 else n) + (i % 2)      # Y: k0 x n (or n x k0 when transposed)
        lda1 = (m1 if ta1 else k1) + (i % 2)     # Z: m1 x k1 (or k1 x m1)
        ey = (n - 1) * ldb0 + k0 if tb0 else (k0 - 1) * ldb0 + n
        ez = (k1 - 1) * lda1 + m1 if ta1 else (m1 - 1) * lda1 + k1
        x_off, y_off = psi_len, arena_len
        psi_len += (k1 - 1) * lda0 + k0
        arena_len += ey
        z_off = arena_len
        arena_len += ez
        recs.append((k1, n, k0, lda0, ldb0, m1, n, k1, lda1, n, 0, tb0, ta1, 0, 0, rng.standard_normal(), rng.standard_normal(),
                     x_off, y_off, z_off, sigma_len))
        sigma_len += m1 * n
    p = np.zeros(len(recs), PAIR_DTYPE)
    for i, r in enumerate(recs):
        p[i] = r
    psi, arena_h = rng.standard_normal(psi_len), rng.standard_normal(arena_len)
    ref = np.zeros(sigma_len)
    oracle.replay(p, arena_h, psi, ref, 0.5, 4)
    pf = type("PF", (), dict(pairs=p, psi_len=psi_len, sigma_len=sigma_len, arena=arena_h, psi=psi))
    for kw in (dict(), dict(keep_order=1), dict(two_stage=1, item_macs=30000)):
        sig, st = _run(gpu, pf, 0.5, **kw)
        assert st["fallback"] == 0 and st["macs_issued"] >= st["macs_executed"] > 0
        assert _close(sig, ref), kw


@pytest.mark.parametrize("seed", [1, 4, 6])
def test_distributive_law_sum_pass_vs_oracle(gpu, seed):
    """the sum pass between the stages (pairs of one psi' sector that multiply the SAME operator block into the SAME window
    first sum their scaled stage-0 products, S = sum alpha_i W_i, then take ONE stage-1 product; b2x_plan.cpp "6b merge
    groups") pinned DIRECTLY against the pair-by-pair oracle: plans with many terms per left operator and blocks wide enough
    for the rewrite to qualify, b2x_plan_stats.n_merged_groups > 0 asserted, 1e-12; keep_order = 1 (none of the rewrites)
    gives the same sigma"""
    rng = np.random.default_rng(900 + seed)
    pf = fill_plan(synth.operator_product_plan(rng, n_row=2, n_col=3, max_dim=420, n_left=2, n_right=6, n_terms=14), seed)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 8)
    sig, st = _run(gpu, pf)
    assert st["n_merged_groups"] >= 20 and st["n_merged_members"] > 2 * st["n_merged_groups"], st
    assert st["n_shared_products"] > 0 and st["n_flipped"] > 0 and st["fallback"] == 0
    assert _close(sig, ref), st
    plain, st1 = _run(gpu, pf, keep_order=1)
    assert st1["n_merged_groups"] == st1["n_shared_products"] == st1["n_flipped"] == 0
    assert _close(plain, ref)
    assert st["macs_executed"] < 0.5 * st1["macs_executed"]  # the rewrites, not rounding luck, are what was tested
