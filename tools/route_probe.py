#!/usr/bin/env python3
"""fused wave path vs two-stage grouped-GEMM path on small plans (golden plans and small structures): ms per H.psi"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from block2_preview_amd import capi, synth
from block2_preview_amd.planfile import read_plan, read_struct_npz
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
capi.device_init(0)
cases = [(os.path.basename(f), read_plan(f)) for f in sorted(glob.glob(os.path.join(root, "tests/golden/*.plan"))) if ".r0of" not in f and ".r1of" not in f and "rot_" not in f]
for f in ("h10_sz_m500_sw1_site4.struct.npz", "cr2_su2_m250_sw1_site5.struct.npz", "cr2_su2_m250_sw1_site30.struct.npz"):
    cases.append((f, read_struct_npz(os.path.join(root, "tests/golden", f))))
h = read_plan(os.path.join(root, "tests/golden/h10szm50.sw1.site5.plan"))
for sc in (2, 4):
    cases.append(("h10szm50.sw1.site5 x%d" % sc, synth.scale_plan(h, sc)))
rng = np.random.default_rng(0)
for name, pf in cases:
    arena = capi.Arena.from_host([rng.random(pf.arena_len)])
    psi = capi.DeviceBuffer(pf.psi_len, rng.random(pf.psi_len))
    sig = capi.DeviceBuffer(pf.sigma_len)
    res = []
    for ts in (-1, 1):
        plan = capi.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len, two_stage=ts)
        for _ in range(5):
            plan.execute_device(psi.ptr, sig.ptr, 1.0)
        capi.device_sync()
        R = 200
        t0 = time.perf_counter()
        for _ in range(R):
            plan.execute_device(psi.ptr, sig.ptr, 1.0)
        capi.device_sync()
        res.append((time.perf_counter() - t0) / R * 1e3)
        plan.close()
    print("%-42s pairs %6d  %8.3f MMAC  fused %.4f ms  two-stage %.4f ms  ratio %.2f" % (name, len(pf.pairs), pf.macs / 1e6, res[0], res[1], res[0] / res[1]), flush=True)
    arena.close()
