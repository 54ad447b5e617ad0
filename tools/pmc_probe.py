#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: (1) a calibration kernel with a known byte count in the SAME
access width the H.psi kernels use (8 B per lane): b2x_vec_axpy on 2 x 1 GiB vectors = 2 GiB read + 1 GiB
written; (2) ONE H.psi of the bench plan (default scale 16 -> M=4000).  The caller sums FETCH_SIZE /
WRITE_SIZE per kernel from the counter CSV (tools/pmc_summary.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from block2_preview_amd import capi, synth
from block2_preview_amd.planfile import read_struct_npz

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # the workload table

arg = sys.argv[1] if len(sys.argv) > 1 else "cr2_m4000"
if arg.isdigit():  # (round-1 form: the Cr2 structure with this multiplier)
    sfile, scale = "cr2_su2_m250_sw1_site20.struct.npz", int(arg)
else:
    sfile, scale = bench.WORKLOADS[arg][0], bench.WORKLOADS[arg][1]
capi.device_init(0)
dev = torch.device("cuda", 0)
n = 1 << 27
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.rand(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
capi.check(capi.lib().b2x_vec_axpy(C.c_double(0.5), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_size_t(n), None))
torch.cuda.synchronize()
del x, y
base = read_struct_npz(os.path.join(root, "tests", "golden", sfile))
full = synth.scale_plan(base, scale) if scale != 1 else base
arena_t = torch.empty(full.arena_len, dtype=torch.float64, device=dev)
for a in range(0, full.arena_len, 1 << 28):
    arena_t[a:a + (1 << 28)].uniform_(0.0, 1.0)
psi = torch.rand(full.psi_len, dtype=torch.float64, device=dev)
sig = torch.zeros(full.sigma_len, dtype=torch.float64, device=dev)
arena = capi.Arena.adopt_device(arena_t.data_ptr(), full.arena_len, keep=arena_t)
plan = capi.Plan(arena, full.pairs, full.psi_len, full.sigma_len)
torch.cuda.synchronize()
plan.execute_device(psi.data_ptr(), sig.data_ptr(), 1.0, 0)
torch.cuda.synchronize()
st = plan.stats
print("PMC_PROBE workload=%s macs=%d op_bytes=%d psi_bytes=%d scale=%d macs_executed=%d macs_issued=%d (SQ_INSTS_MFMA of the gg_kernel launches x 1024 should equal macs_issued)"
      % (arg, st["macs"], st["op_elems_unique"] * 8, full.psi_len * 8, scale, st["macs_executed"], st["macs_issued"]))
