"""Per-iteration cost of the device-resident Davidson next to its H.psi: tools/davidson_overhead.py <workload> [iters].
The operator data are random (H is not symmetric: the eigenvalue means nothing), soft_max_iter fixes the iteration count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from block2_preview_amd import capi, synth, b2x_host
from block2_preview_amd.planfile import read_struct_npz

w = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fn, scale, M, _ = bench.WORKLOADS[w]
pf = read_struct_npz(os.path.join(bench.GOLD, fn))
if scale > 1:
    pf = synth.scale_plan(pf, scale)
capi.device_init(0)
b2x_host.device_init(0)
dev = torch.device("cuda:0")
a = torch.rand(pf.arena_len, dtype=torch.float64, device=dev)
arena = capi.Arena.adopt_device(a.data_ptr(), pf.arena_len, keep=a)
plan = capi.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len)
n = pf.psi_len
psi = torch.rand(n, dtype=torch.float64, device=dev)
sig = torch.zeros(n, dtype=torch.float64, device=dev)
diag = torch.rand(n, dtype=torch.float64, device=dev) + 1.0
for _ in range(3):
    plan.execute_device(psi.data_ptr(), sig.data_ptr(), 1.0, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    plan.execute_device(psi.data_ptr(), sig.data_ptr(), 1.0, 0)
torch.cuda.synchronize()
t_h = (time.perf_counter() - t0) / iters
for rep in range(2):
    ket = psi.clone()
    t0 = time.perf_counter()
    e, nd = b2x_host.davidson_device(plan._h.value, diag.data_ptr(), ket.data_ptr(), n, 1e-30, 5000, iters)
    t_d = (time.perf_counter() - t0) / max(nd, 1)
print("%s: psi %d, H.psi %.3f ms, Davidson iteration %.3f ms (%d iterations): overhead %.3f ms = %.0f %% of H.psi"
      % (w, n, t_h * 1e3, t_d * 1e3, nd, (t_d - t_h) * 1e3, 100 * (t_d - t_h) / t_h))
