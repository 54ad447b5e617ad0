"""Repeated device-pointer execution with CHANGING psi / sigma / scale on every route of the plan executor.

b2x_plan_execute(on_device=1) replays the launches of one H.psi from a HIP graph whose kernel nodes get their psi, sigma
and scale arguments patched per call (csrc/b2x_capi.cpp run_plan_graph).  Which argument of a kernel is psi is recorded by
the launcher of that kernel (psi is argument 3 of gg_kernel, argument 4 of hpsi_wave): a replay that patched the wrong slot
would silently compute H.psi_old — what every Davidson iteration after the first would see.  Each case below makes three
calls with three different (psi, sigma, scale) buffer triples, then destroys the plan, creates it again on ANOTHER arena
(a compiled-plan cache hit, re-bound) and executes through the graph once more; every result is compared with the CPU
oracle.  Run with the graph on (the product default) and off (direct launches)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, fill_plan
from block2_preview_amd import synth
from block2_preview_amd.planfile import read_gemm_list
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-12

ROUTES = {
    "default": {},
    "fused_wave": {"two_stage": -1},          # hpsi_wave, class chosen by the compiler
    "fused_16": {"tile_n": 16},               # forced fused classes
    "fused_32": {"tile_n": 32},
    "fused_64": {"tile_n": 64},
    "two_stage": {"two_stage": 1},
    "keep_order": {"keep_order": 1},
    "generic": {"kernel": 1},
}


def _three_calls(gpu, plan, n_in, n_out, reference, seed):
    rng = np.random.default_rng(seed)
    bufs = []
    for call, scale in enumerate((1.0, -0.5, 2.25)):
        x, s0 = rng.random(n_in), rng.random(n_out)
        dx, ds = gpu.DeviceBuffer(n_in, x), gpu.DeviceBuffer(n_out, s0)
        bufs.append((dx, ds))  # (kept alive: every call sees fresh addresses)
        plan.execute_device(dx.ptr, ds.ptr, scale)
        gpu.device_sync()
        ref = s0.copy()
        reference(x, ref, scale)
        got = ds.download()
        assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max()), "call %d (scale %g)" % (call, scale)
    # the first buffers again (graph patched back)
    dx, ds = bufs[0]
    x, before = dx.download(), ds.download()
    plan.execute_device(dx.ptr, ds.ptr, 1.0)
    gpu.device_sync()
    ref = before.copy()
    reference(x, ref, 1.0)
    assert np.abs(ds.download() - ref).max() <= TOL * max(1.0, np.abs(ref).max())
    for dx, ds in bufs:
        dx.close(), ds.close()


@pytest.mark.parametrize("graph", ["1", "0"])
@pytest.mark.parametrize("route", list(ROUTES))
def test_pair_plan_routes(gpu, monkeypatch, route, graph):
    monkeypatch.setenv("B2X_GRAPH", graph)
    rng = np.random.default_rng(2024)
    pf = fill_plan(synth.random_rotate_plan(rng, n_sectors=4, max_dim=150, max_terms=8), 5)
    gpu.plan_cache_clear()
    for rebind, arena_seed in enumerate((5, 6)):  # second round: same records on another arena -> cache hit, re-bound
        data = np.random.default_rng(arena_seed).random(pf.arena_len)
        arena = gpu.Arena.from_host([data])
        hits0 = gpu.plan_cache_stats()[0]
        plan = gpu.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len, **ROUTES[route])
        if rebind and route != "generic":
            assert gpu.plan_cache_stats()[0] == hits0 + 1, "the re-created plan must come from the cache"
        st = plan.stats
        if route == "fused_wave":
            assert st["macs_issued"] == 0, "the fused wave kernel was asked for"
        _three_calls(gpu, plan, pf.psi_len, pf.sigma_len,
                     lambda x, out, sc: oracle.replay(pf.pairs, data, x, out, sc), 100 + rebind)
        plan.close(), arena.close()
    gpu.plan_cache_clear()


@pytest.mark.parametrize("graph", ["1", "0"])
@pytest.mark.parametrize("name", ["p_n2su2.sw1.site5.pnoise", "p_h10sz.sw1.site4.pnoise"])
def test_gemm_list_plan(gpu, monkeypatch, name, graph):
    monkeypatch.setenv("B2X_GRAPH", graph)
    gl = read_gemm_list(os.path.join(GOLDEN, name))
    gpu.plan_cache_clear()
    for rebind, data in enumerate((gl.arena, np.random.default_rng(9).random(gl.arena.size))):
        arena = gpu.Arena.from_host([data])
        plan = gpu.GemmPlan(arena, gl.gemms, gl.in_len, gl.out_len)
        _three_calls(gpu, plan, gl.in_len, gl.out_len,
                     lambda x, out, sc: oracle.gemm_list(gl.gemms, data, x, out, sc), 200 + rebind)
        plan.close(), arena.close()
    gpu.plan_cache_clear()
