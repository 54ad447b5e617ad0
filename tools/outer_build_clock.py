import os, sys
sys.path.insert(0, "/root/repo")
from block2_preview_amd import capi
from block2_preview_amd.sweep import DMRG, ChainFixture
capi.device_init(0)
fx = ChainFixture("tests/golden/chain_cr2_m250_cut9/cr2g").preload()
dm = DMRG(fx, "su2", conv_thrd=1e-18)
dm.init_environments()
for isw in range(3):
    dm.sweep(isw, isw % 2 == 0)
    print("SWEEP", isw, file=sys.stderr)
