"""Host cost of a plan: the plan compiler alone (test hook, no device) next to b2x_plan_create (compile + device allocation
+ upload of the work lists).  tools/compile_time.py <struct.npz> <scale>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hooks
from block2_preview_amd import capi, synth
from block2_preview_amd.planfile import read_struct_npz

pf = read_struct_npz(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
f = int(sys.argv[2])
if f > 1:
    pf = synth.scale_plan(pf, f)
t_c = 1e9
for i in range(7):  # (minimum of 7: the container's cores are shared)
    t = time.perf_counter()
    st, fb = hooks.debug_compile_and_emulate(pf.pairs, pf.psi_len, pf.sigma_len, None, None, None, arena_len=pf.arena_len)
    t_c = min(t_c, time.perf_counter() - t)
import torch
if torch.cuda.is_available():
    capi.device_init(0)
    a = torch.empty(pf.arena_len, dtype=torch.float64, device="cuda")
    arena = capi.Arena.adopt_device(a.data_ptr(), pf.arena_len, keep=a)
    for i in range(2):
        t = time.perf_counter()
        plan = capi.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len)
        t_p = time.perf_counter() - t
        plan.close()
    print("%s x%d: compile alone %.1f ms, b2x_plan_create %.1f ms, items %d, device MB %.0f" % (sys.argv[1], f, t_c * 1e3, t_p * 1e3, st["n_items"], st["device_bytes"] / 1e6))
else:
    print("%s x%d: compile alone %.1f ms, items %d" % (sys.argv[1], f, t_c * 1e3, st["n_items"]))
