#!/usr/bin/env python3
"""Kernel probe: uniform synthetic plans (every pair d x d x d, one psi' sector) timed through the C ABI.
usage: gg_probe.py [d] [n_pairs] [two_stage] [tb0] [ta1]   -> prints TFLOP/s of the H.psi replay"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from block2_preview_amd import capi
from block2_preview_amd.planfile import PAIR_DTYPE

d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
two = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tb0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ta1 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
capi.device_init(0)
p = np.zeros(n, PAIR_DTYPE)
for nm in ("m0", "n0", "k0", "lda0", "ldb0", "m1", "n1", "k1", "lda1", "ldc1"):
    p[nm] = d
p["tb0"], p["ta1"], p["alpha0"], p["alpha1"] = tb0, ta1, 1.0, 1.0
nsec = 8  # psi / psi' sectors
p["x_off"] = (np.arange(n) % nsec) * d * d
p["v_off"] = ((np.arange(n) // 2) % nsec) * d * d
p["y_off"] = np.arange(n) * 2 * d * d
p["z_off"] = np.arange(n) * 2 * d * d + d * d
dev = torch.device("cuda", 0)
if os.environ.get("B2X_PROBE_DATA") == "ones":  # power probe: constant operands toggle few bits in the MFMA pipe
    arena_t = torch.ones(2 * n * d * d, dtype=torch.float64, device=dev)
    psi = torch.ones(nsec * d * d, dtype=torch.float64, device=dev)
elif os.environ.get("B2X_PROBE_DATA") == "zeros":
    arena_t = torch.zeros(2 * n * d * d, dtype=torch.float64, device=dev)
    psi = torch.zeros(nsec * d * d, dtype=torch.float64, device=dev)
else:
    arena_t = torch.rand(2 * n * d * d, dtype=torch.float64, device=dev)
    psi = torch.rand(nsec * d * d, dtype=torch.float64, device=dev)
sig = torch.zeros(nsec * d * d, dtype=torch.float64, device=dev)
arena = capi.Arena.adopt_device(arena_t.data_ptr(), arena_t.numel(), keep=arena_t)
plan = capi.Plan(arena, p, psi.numel(), sig.numel(), two_stage=two, item_macs=int(float(os.environ.get('B2X_ITEM_MACS', '0'))))
st = plan.stats
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    plan.execute_device(psi.data_ptr(), sig.data_ptr(), 1.0, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    plan.execute_device(psi.data_ptr(), sig.data_ptr(), 1.0, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
k_ms, tot_ms = plan.time_kernel(psi.data_ptr(), sig.data_ptr(), 3, s)
print("d=%d n=%d two_stage=%d tb0=%d ta1=%d: %.2f TFLOP/s wall (%.2f ms), gemm kernels %.2f TFLOP/s (%.2f ms), items %d steps-> exec/alg %.3f"
      % (d, n, two, tb0, ta1, 2 * st["macs"] / dt / 1e12, dt * 1e3, 2 * st["macs_alg_dominant"] / (k_ms * 1e-3) / 1e12, k_ms,
         st["n_items"], st["macs_executed"] / st["macs"]))
