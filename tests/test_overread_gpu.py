"""The kernels fetch operands as 16-byte granules but must never touch anything outside the caller's buffers, nor
consume anything a kernel has not written: psi, the arena and the input vector of a GEMM list are placed between NaN guard
bands and handed over with their EXACT length (no slack), and under B2X_DEBUG_POISON=1 the library fills everything it
allocates itself — the slack behind owned buffers, the whole W scratch before its first use (except the padding element
behind every slot, which production keeps zero) — with NaN.  A stray read multiplies a NaN into the result (0 * NaN =
NaN), so equality with the oracle proves there is none.  Shapes: odd dimensions, K not a multiple of 16, both operand
layouts, operands that end exactly at the end of a buffer (those are staged, b2x_plan_stats.n_staged)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, fill_plan
from block2_preview_amd import synth
from block2_preview_amd.planfile import PAIR_DTYPE, read_gemm_list
from oracle import oracle

pytestmark = pytest.mark.gpu
GUARD = 64


def _guarded(gpu, host):
    """device buffer [NaN x GUARD | host | NaN x GUARD]; returns (buffer, device address of the payload)"""
    buf = np.full(len(host) + 2 * GUARD, np.nan)
    buf[GUARD:GUARD + len(host)] = host
    d = gpu.DeviceBuffer(len(buf), buf)
    return d, d.ptr + 8 * GUARD


def _run_guarded(gpu, pf, **kw):
    ar_buf, ar_ptr = _guarded(gpu, pf.arena)
    psi_buf, psi_ptr = _guarded(gpu, pf.psi)
    sig_buf, sig_ptr = _guarded(gpu, np.zeros(pf.sigma_len))
    arena = gpu.Arena.adopt_device(ar_ptr, pf.arena_len, keep=ar_buf)
    plan = gpu.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len, **kw)
    plan.execute_device(psi_ptr, sig_ptr, 1.0)
    gpu.device_sync()
    out = sig_buf.download()
    st = plan.stats
    plan.close(), arena.close()
    assert np.isnan(out[:GUARD]).all() and np.isnan(out[-GUARD:]).all()  # nothing written outside sigma either
    return out[GUARD:-GUARD], st


def _check(gpu, pf, **kw):
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 4)
    sig, st = _run_guarded(gpu, pf, **kw)
    assert np.isfinite(sig).all(), st
    assert np.abs(sig - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), st
    return st


@pytest.mark.parametrize("poison", ["0", "1"])
@pytest.mark.parametrize("seed,max_dim", [(0, 3), (1, 17), (2, 33), (3, 75), (4, 150)])
def test_random_plans_exact_buffers(gpu, monkeypatch, poison, seed, max_dim):
    monkeypatch.setenv("B2X_DEBUG_POISON", poison)
    rng = np.random.default_rng(4000 + seed)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=4, max_dim=max_dim, max_terms=8), seed)
    _check(gpu, pf, two_stage=1, keep_order=1)  # grouped-GEMM kernel, the reference's order (every K a k0 or k1 of a pair)
    _check(gpu, pf, two_stage=1)                # ... with re-association (W' = op(Z).X as the k-contiguous A operand)
    _check(gpu, pf)                             # default routing (small plans: the fused wave kernel)


@pytest.mark.parametrize("poison", ["0", "1"])
@pytest.mark.parametrize("seed", range(3))
def test_shared_products_exact_buffers(gpu, monkeypatch, poison, seed):
    monkeypatch.setenv("B2X_DEBUG_POISON", poison)
    rng = np.random.default_rng(4100 + seed)
    pf = fill_plan(synth.operator_product_plan(rng, n_row=3, n_col=3, max_dim=[9, 45, 131][seed], n_left=3, n_right=3,
                                               n_terms=8), seed)
    _check(gpu, pf, two_stage=1)
    _check(gpu, pf, two_stage=1, scratch_mb=1)


@pytest.mark.parametrize("poison", ["0", "1"])
def test_degenerate_operands_at_buffer_ends_are_staged(gpu, monkeypatch, poison):
    """K = 1 (k-contiguous A) and one-row (row-contiguous A) operands whose 16-byte fetch would touch the element behind
    the buffer: X = the last column vector of psi (k0 = 1), Z = a one-column block read transposed (m1 = 1, lda1 = 3) that
    ends the arena.  The plan stages them in its own memory (n_staged) and the result is the oracle's."""
    monkeypatch.setenv("B2X_DEBUG_POISON", poison)
    rng = np.random.default_rng(77)
    p = np.zeros(2, PAIR_DTYPE)
    psi_len, arena_len, sigma_len = 50, 400, 40
    # pair 0: X (7 x 1, lda 1) = the last 7 elements of psi; Y (1 x 5); Z (6 x 7); V 6 x 5
    p[0] = (7, 5, 1, 1, 5, 6, 5, 7, 7, 5, 0, 0, 0, 0, 0, 1.0, 0.5, psi_len - 7, 0, 10, 0)
    # pair 1: X (9 x 4); Y (4 x 5); Z read transposed: stored 9 x 1 with lda1 = 3 (m1 = 1), ending exactly at the arena end
    z_off = arena_len - ((9 - 1) * 3 + 1)
    p[1] = (9, 5, 4, 4, 5, 1, 5, 9, 3, 5, 0, 0, 1, 0, 0, 1.0, -1.5, 0, 100, z_off, 30)
    pf = synth.random_rotate_plan(rng, 1, 2, 1)  # (container only)
    pf.pairs, pf.psi_len, pf.sigma_len, pf.arena_len = p, psi_len, sigma_len, arena_len
    pf.arena, pf.psi = rng.random(arena_len), rng.random(psi_len)
    st = _check(gpu, pf, two_stage=1, keep_order=1)
    assert st["n_staged"] >= 2, st
    _check(gpu, pf, two_stage=1)
    _check(gpu, pf)


@pytest.mark.parametrize("poison", ["0", "1"])
@pytest.mark.parametrize("name", ["p_n2su2.sw1.site5.pnoise", "p_h10sz.sw1.site4.pnoise"])
def test_gemm_list_exact_buffers(gpu, monkeypatch, poison, name):
    monkeypatch.setenv("B2X_DEBUG_POISON", poison)
    gl = read_gemm_list(os.path.join(GOLDEN, name))
    ar_buf, ar_ptr = _guarded(gpu, gl.arena)
    in_buf, in_ptr = _guarded(gpu, gl.vin)
    out_buf, out_ptr = _guarded(gpu, np.zeros(gl.out_len))
    arena = gpu.Arena.adopt_device(ar_ptr, gl.arena_len, keep=ar_buf)
    for keep in (0, 1):
        out_buf.upload(np.concatenate([np.full(GUARD, np.nan), np.zeros(gl.out_len), np.full(GUARD, np.nan)]))
        plan = gpu.GemmPlan(arena, gl.gemms, gl.in_len, gl.out_len, keep_order=keep)
        plan.execute_device(in_ptr, out_ptr, 1.0)
        gpu.device_sync()
        out = out_buf.download()[GUARD:-GUARD]
        plan.close()
        assert np.abs(out - gl.out_ref).max() <= 1e-12 * max(1.0, np.abs(gl.out_ref).max())
    arena.close()
