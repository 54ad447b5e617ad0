"""The plan COMPILER (pairs -> tiles/parts/items, block2-preview_amd/csrc/b2x_plan.cpp) checked on the CPU:
the compiled work list, evaluated with plain host loops through the test hook
b2x_debug_compile_and_emulate, must reproduce the oracle.  (The HIP kernels that consume the same work
list are checked against the oracle in test_hpsi_gpu.py.)"""
import os

import numpy as np
import pytest

import hooks
from conftest import fill_plan, golden_plan_files
from block2_preview_amd import capi, synth
from block2_preview_amd.planfile import PAIR_DTYPE, read_plan
from oracle import oracle

FILES = golden_plan_files()


def _check(pf, **kw):
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 0.75)
    sig = np.zeros(pf.sigma_len)
    st, fb = hooks.debug_compile_and_emulate(pf.pairs, pf.psi_len, pf.sigma_len, pf.arena, pf.psi, sig, 0.75, **kw)
    assert not fb
    assert st["macs"] == pf.macs
    assert np.abs(sig - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    return st


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_compiled_golden(built, fn):
    _check(read_plan(fn))


@pytest.mark.parametrize("seed", range(6))
def test_compiled_random_with_slices(built, seed):
    rng = np.random.default_rng(100 + seed)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=4, max_dim=90, max_terms=6), seed)
    st = _check(pf)
    assert st["n_targets"] <= 4  # overlapping row/col slices merge into their sector


@pytest.mark.parametrize("tile_n,item_macs", [(16, 1), (32, 50000), (64, 0), (128, 1 << 40)])
def test_compiled_forced_classes_and_item_sizes(built, tile_n, item_macs):
    """every kernel class / split granularity yields the same sum (k1 chunking, multi-tile, multi-item)"""
    rng = np.random.default_rng(5)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=3, max_dim=150, max_terms=4), 5)
    _check(pf, tile_n=tile_n, item_macs=item_macs)


def test_scaled_structure_consistent(built):
    """scale_plan keeps the plan well-formed (what bench.py feeds the GPU at M=4000 scale)"""
    pf = read_plan([f for f in FILES if "h10szm50.sw0.site6" in f][0])  # has column slices
    big = fill_plan(synth.scale_plan(pf, 2), 3)
    assert big.macs == 8 * pf.macs
    _check(big)


def test_empty_and_invalid(built):
    st, fb = hooks.debug_compile_and_emulate(np.zeros(0, PAIR_DTYPE), 10, 10, np.zeros(4), np.zeros(10), np.zeros(10))
    assert st["n_pairs"] == 0 and not fb
    bad = np.zeros(1, PAIR_DTYPE)
    bad["m0"] = bad["n0"] = bad["k0"] = bad["m1"] = bad["n1"] = bad["k1"] = 4
    bad["lda0"] = bad["ldb0"] = bad["lda1"] = bad["ldc1"] = 4
    bad["x_off"] = 100  # runs past psi
    with pytest.raises(capi.B2XError):
        hooks.debug_compile_and_emulate(bad, 16, 16, np.zeros(32), np.zeros(16), np.zeros(16))
    bad["x_off"] = 0
    bad["ta0"] = 1  # unsupported on this path
    with pytest.raises(capi.B2XError):
        hooks.debug_compile_and_emulate(bad, 16, 16, np.zeros(32), np.zeros(16), np.zeros(16))


@pytest.mark.parametrize("fn", FILES[:3], ids=[os.path.basename(f) for f in FILES[:3]])
def test_compiled_two_stage_golden(built, fn):
    st = _check(read_plan(fn), two_stage=1, keep_order=1)
    assert st["macs_issued"] > 0 and st["macs_executed"] == st["macs"]  # the reference's order: no recomputation
    st = _check(read_plan(fn), two_stage=1)  # per pair the cheaper association: same result, never more work
    assert st["macs_executed"] <= st["macs"] and st["macs_alg_dominant"] == st["macs"]


@pytest.mark.parametrize("scratch_mb,item_macs", [(1, 0), (1, 100000), (3, 1 << 40)])
def test_compiled_two_stage_multi_superstep(built, scratch_mb, item_macs):
    rng = np.random.default_rng(17)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=3, max_dim=400, max_terms=4), 17)
    _check(pf, two_stage=1, scratch_mb=scratch_mb, item_macs=item_macs)


@pytest.mark.parametrize("seed,scratch_mb,keep_order", [(0, 0, 0), (1, 0, 1), (2, 1, 0), (3, 1, 0), (4, 0, 0)])
def test_shared_products_and_association(built, seed, scratch_mb, keep_order):
    """H = sum of (left operator) x (right operator) products: pairs share stage-0 products and take either association;
    with a 1 MiB scratch the sharing groups are split over super-steps.  Same sigma as the pair-by-pair oracle, and
    strictly fewer MACs executed than the reference's order needs."""
    rng = np.random.default_rng(400 + seed)
    pf = fill_plan(synth.operator_product_plan(rng, n_row=3, n_col=4, max_dim=[40, 90, 200, 160, 30][seed], n_left=4,
                                               n_right=3, n_terms=9), seed)
    st = _check(pf, two_stage=1, scratch_mb=scratch_mb, keep_order=keep_order)
    if keep_order:
        assert st["macs_executed"] == st["macs"]
    else:
        assert st["macs_executed"] < st["macs"]
    st2 = _check(pf, scratch_mb=scratch_mb, keep_order=keep_order)  # auto routing
    assert st2["macs"] == st["macs"]
    if not keep_order:  # operator pre-sums on top (second operators of pairs sharing a product and a window)
        st3 = _check(pf, two_stage=1, scratch_mb=scratch_mb, presum=1)
        assert st3["macs_executed"] <= st["macs_executed"]


def test_degenerate_operands_at_buffer_ends_are_staged(built):
    """A operands whose 16-byte fetch would touch the element behind psi or the arena (they end exactly at the end of the
    buffer) are read from a staged copy in plan-owned memory (b2x_plan_stats.n_staged); everything else is read in place"""
    rng = np.random.default_rng(77)
    p = np.zeros(2, PAIR_DTYPE)
    psi_len, arena_len, sigma_len = 50, 400, 40
    p[0] = (7, 5, 1, 1, 5, 6, 5, 7, 7, 5, 0, 0, 0, 0, 0, 1.0, 0.5, psi_len - 7, 0, 10, 0)
    z_off = arena_len - ((9 - 1) * 3 + 1)
    p[1] = (9, 5, 4, 4, 5, 1, 5, 9, 3, 5, 0, 0, 1, 0, 0, 1.0, -1.5, 0, 100, z_off, 30)
    pf = synth.random_rotate_plan(rng, 1, 2, 1)
    pf.pairs, pf.psi_len, pf.sigma_len, pf.arena_len = p, psi_len, sigma_len, arena_len
    pf.arena, pf.psi = rng.random(arena_len), rng.random(psi_len)
    st = _check(pf, two_stage=1, keep_order=1)
    assert st["n_staged"] == 2
    p["x_off"][0] -= 1  # one element of room behind the operands: nothing to stage
    p["z_off"][1] -= 1
    st = _check(pf, two_stage=1, keep_order=1)
    assert st["n_staged"] == 0


def _all_structures():
    import glob

    from conftest import GOLDEN

    return sorted(glob.glob(os.path.join(GOLDEN, "*.struct.npz")) + glob.glob(os.path.join(GOLDEN, "*.rotstruct.npz")))


@pytest.mark.parametrize("fn", _all_structures(), ids=os.path.basename)
def test_reference_plans_never_take_the_atomic_fallback(built, fn):
    """every plan structure captured from the reference (N2, H10 M=500, Hubbard M=3000, Cr2 M=250 H.psi and rotations)
    segments by output: the non-deterministic per-pair atomic kernel is never selected; at most the operands that end
    the arena / psi are staged (no slack is assumed behind either buffer)"""
    from block2_preview_amd.planfile import read_struct_npz

    pf = read_struct_npz(fn)
    for kw in ({}, {"keep_order": 1}):
        st, fb = hooks.debug_compile_and_emulate(pf.pairs, pf.psi_len, pf.sigma_len, None, None, None,
                                                 arena_len=pf.arena_len, **kw)
        assert not fb and st["fallback"] == 0 and st["n_staged"] <= 8 and st["macs"] == pf.macs


def test_compilations_are_independent_of_what_was_compiled_before(built):
    """The compiler's temporaries live in a per-thread arena that is rewound, not released, between compilations
    (b2x_plan.cpp ScratchArena): a plan compiled after OTHER plans (larger, smaller, forced options) must give the bits
    it gives when compiled first."""
    rng = np.random.default_rng(77)
    small = fill_plan(synth.random_rotate_plan(rng, n_sectors=3, max_dim=40, max_terms=3), 1)
    large = fill_plan(synth.random_rotate_plan(rng, n_sectors=6, max_dim=200, max_terms=8), 2)

    def run(pf, **kw):
        sig = np.zeros(pf.sigma_len)
        st, fb = hooks.debug_compile_and_emulate(pf.pairs, pf.psi_len, pf.sigma_len, pf.arena, pf.psi, sig, 1.0, **kw)
        assert not fb
        return sig, st

    first, st0 = run(small)
    for pf, kw in [(large, {}), (large, dict(keep_order=1)), (small, dict(scratch_mb=1)), (large, dict(presum=1))]:
        run(pf, **kw)
        again, st1 = run(small)
        assert np.array_equal(first, again) and st0 == st1


def test_sum_pass_groups_are_counted_and_exact(built):
    """b2x_plan_stats counts the three rewrites of the pair list; the compiled work list of a plan WITH merged groups
    (distributive-law sum pass) evaluated by host loops equals the pair-by-pair oracle"""
    import hooks
    from block2_preview_amd import synth
    from oracle import oracle

    rng = np.random.default_rng(904)
    pf = synth.operator_product_plan(rng, n_row=2, n_col=3, max_dim=420, n_left=2, n_right=6, n_terms=14)
    g = np.random.default_rng(4)
    pf.arena, pf.psi = g.random(pf.arena_len), g.random(pf.psi_len)
    sig, ref = np.zeros(pf.sigma_len), np.zeros(pf.sigma_len)
    st, fb = hooks.debug_compile_and_emulate(pf.pairs, pf.psi_len, pf.sigma_len, pf.arena, pf.psi, sig)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 4)
    assert not fb and st["n_merged_groups"] >= 20 and st["n_merged_members"] > 2 * st["n_merged_groups"]
    assert st["n_shared_products"] > 0 and st["n_flipped"] > 0
    assert np.abs(sig - ref).max() <= 1e-12 * np.abs(ref).max()
    sig1 = np.zeros(pf.sigma_len)
    st1, _ = hooks.debug_compile_and_emulate(pf.pairs, pf.psi_len, pf.sigma_len, pf.arena, pf.psi, sig1, keep_order=1)
    assert st1["n_merged_groups"] == st1["n_shared_products"] == st1["n_flipped"] == 0
    assert np.abs(sig1 - ref).max() <= 1e-12 * np.abs(ref).max()
