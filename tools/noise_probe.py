#!/usr/bin/env python3
"""Perturbative-noise probe: a Cr2/SVP noise GEMM list (structure captured from the reference, tests/golden/
*.pnoise_struct.npz) scaled to bond dimension 250 x scale, synthetic data, timed through the C ABI.
usage: noise_probe.py [struct.npz] [scale ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from block2_preview_amd import capi, synth
from block2_preview_amd.planfile import read_gemm_list

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fn = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "tests/golden/cr2_su2_m250_sw1_site20.pnoise_struct.npz")
scales = [int(x) for x in sys.argv[2:]] or [1, 4, 16]
capi.device_init(0)
dev = torch.device("cuda", 0)
base = read_gemm_list(fn)
for f in scales:
    gl = synth.scale_gemm_list(base, f)
    arena_t = torch.rand(gl.arena_len + 8, dtype=torch.float64, device=dev)
    vin = torch.rand(gl.in_len + 8, dtype=torch.float64, device=dev)
    out = torch.zeros(gl.out_len, dtype=torch.float64, device=dev)
    arena = capi.Arena.adopt_device(arena_t.data_ptr(), gl.arena_len, keep=arena_t)
    t0 = time.perf_counter()
    plan = capi.GemmPlan(arena, gl.gemms, gl.in_len, gl.out_len)
    t_compile = time.perf_counter() - t0
    st = plan.stats
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.execute_device(vin.data_ptr(), out.data_ptr(), 1.0, s)
    torch.cuda.synchronize()
    R = 5
    t0 = time.perf_counter()
    for _ in range(R):
        plan.execute_device(vin.data_ptr(), out.data_ptr(), 1.0, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    k_ms, _ = plan.time_kernel(vin.data_ptr(), out.data_ptr(), 3, s)
    print("M=%d: %d gemms, %.3f TMAC, out %.1f M doubles, operators %.1f GB | %.2f ms/replay = %.2f TFLOP/s (gg kernel %.2f ms "
          "= %.2f TFLOP/s), useful/issued %.3f, items %d, compile %.2f s"
          % (250 * f, len(gl.gemms), gl.macs / 1e12, gl.out_len / 1e6, gl.arena_len * 8 / 1e9, dt * 1e3,
             2 * gl.macs / dt / 1e12, k_ms, 2 * gl.macs / (k_ms * 1e-3) / 1e12, gl.macs / max(1, st["macs_issued"]),
             st["n_items"], t_compile), flush=True)
    plan.close(), arena.close()
    del arena_t, vin, out
    torch.cuda.empty_cache()
