#!/usr/bin/env python3
"""cProfile of one chain replay (host side of the sweep loop): usage sweep_profile.py <prefix> <su2|sz> <n_sweeps> [first profiled sweep]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from block2_preview_amd import capi  # noqa: E402
from block2_preview_amd.sweep import DMRG, ChainFixture  # noqa: E402

prefix, sym, n_sw = sys.argv[1], sys.argv[2], int(sys.argv[3])
capi.device_init(0)
fx = ChainFixture(prefix).preload()
dm = DMRG(fx, sym)
dm.init_environments()
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # sweeps before this one run outside the profile
for isw in range(first):
    dm.sweep(isw, isw % 2 == 0)
pr = cProfile.Profile()
pr.enable()
for isw in range(first, n_sw):
    dm.sweep(isw, isw % 2 == 0)
pr.disable()
print({k: round(v, 3) for k, v in dm.tm.items()})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
