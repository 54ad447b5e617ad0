#!/usr/bin/env python3
"""Blocking probe (b2x_outer_build on device-resident vectors):
  real   — the Cr2/SVP M=250 blocking term list captured from the reference (tests/golden/*.blkstruct.npz), synthetic data
  synth  — S enlarged sectors of g x g sub-blocks (b x b each), T block-times-scalar terms per sub-block: the shape the
           real list takes at bond dimension ~ g*b, without its small-block tail
usage: blk_probe.py real [blkstruct.npz]   |   blk_probe.py synth b [g] [T] [S]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from block2_preview_amd import capi
from block2_preview_amd.planfile import OUTER_TERM_DTYPE, read_outer_struct_npz

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "real"
capi.device_init(0)
dev = torch.device("cuda", 0)
if mode == "real":
    fn = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "tests/golden/cr2_su2_m250_sw0_c20_lblk.blkstruct.npz")
    t, lens = read_outer_struct_npz(fn)
    arena_len, in_len, out_len = int(lens[1]), int(lens[2]), int(lens[3])
    label = os.path.basename(fn)
else:
    b = int(sys.argv[2])
    g = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    S = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    n_blocks = 64
    arena_len, in_len, out_len = 16, n_blocks * b * b, S * (g * b) ** 2
    rows = []
    rng = np.random.default_rng(0)
    for s in range(S):
        for i in range(g):
            for j in range(g):
                for k in range(T):
                    blk = int(rng.integers(n_blocks))
                    tr = k % 3 == 2  # a third of the terms read the block transposed
                    rows.append((b, b, 1 if tr else b, b if tr else 1, 0, 0, g * b, 1, 0, (0, 0), 0.5 + k,
                                 blk * b * b, int(rng.integers(16)), s * (g * b) ** 2 + i * b * g * b + j * b))
    t = np.array(rows, OUTER_TERM_DTYPE)
    label = "synthetic b=%d g=%d T=%d S=%d" % (b, g, T, S)
arena_t = torch.rand(arena_len + 8, dtype=torch.float64, device=dev)
vin = torch.rand(in_len + 8, dtype=torch.float64, device=dev)
out = torch.zeros(out_len, dtype=torch.float64, device=dev)
arena = capi.Arena.adopt_device(arena_t.data_ptr(), arena_len, keep=arena_t)
elems = int((t["m"].astype(np.int64) * t["n"]).sum())
for _ in range(2):
    capi.outer_build(arena, t, vin.data_ptr(), out.data_ptr(), True, in_len, out_len)
R = 3
t1 = time.perf_counter()
for _ in range(R):
    capi.outer_build(arena, t, vin.data_ptr(), out.data_ptr(), True, in_len, out_len)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / R
alg = 8 * (elems + 2 * out_len)  # every term reads its block once; the output is read and written once
print("%s: %d terms, %.1f M term elements, out %.1f M doubles, %.2f GB algorithmic | %.2f ms per call (host compile + upload "
      "+ kernel)" % (label, len(t), elems / 1e6, out_len / 1e6, alg / 1e9, dt * 1e3), flush=True)
