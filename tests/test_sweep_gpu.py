"""The energy gate: the repo's own two-site sweep loop (block2-preview_amd/sweep.py), every operator block device-resident
from the initial environments on, over the symbolic fixtures of ONE reference run (N2/STO-3G, SU2, M=200, no noise;
tests/golden/chain_n2su2: 109 blocking / rotation / operator-sum / effective-Hamiltonian events, no operator or
wavefunction data).
Every site energy of the forward sweep 0 and of the backward sweep 1 must equal the reference's to 1e-7 Ha, and the last
one the known answer of block2's own test, E(N2/STO-3G) = -107.654122447525 (unit_test/test_dmrg_n2_sto3g.cpp:187)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_n2_su2_two_sweeps_site_energies(gpu):
    from block2_preview_amd.sweep import DMRG, ChainFixture

    fx = ChainFixture(os.path.join(GOLDEN, "chain_n2su2", "n2c"))
    assert len(fx.events) == 109 and len(fx.ref_energy) == 18
    dm = DMRG(fx, "su2")
    blocks = dm.init_environments()
    assert dm.n_sites == 10 and blocks == list(range(2, 10))
    e0 = dm.sweep(0, True)
    e1 = dm.sweep(1, False)
    assert fx.pos == len(fx.events)  # every step the reference did was done here, in its order
    worst = 0.0
    for (isw, site), ref in fx.ref_energy.items():
        worst = max(worst, abs(dm.energies[(isw, site)] - ref))
    print("site energies", ["%.10f" % e for e in e0 + e1], "worst |dE| = %.2e" % worst, "timers", dict(dm.tm), "zero-filled", dm.zero_log)
    assert worst < 1e-7
    assert abs(e1[-1] - (-107.654122447525)) < 1e-7
    assert abs(e0[0] - (-99.0104099582)) < 1e-7  # far from converged at the first site: the chain, not the answer, is tested


def test_h10_sz_two_sweeps_site_energies(gpu):
    """the same gate on an SZ system: H10/STO-6G R=1.8 (the molecule of BASELINE configs[1]) at M=100, sweeps 0-1 of one
    reference run (tests/golden/chain_h10sz: 114 events); every site energy to 1e-7 Ha"""
    from block2_preview_amd.sweep import DMRG, ChainFixture

    fx = ChainFixture(os.path.join(GOLDEN, "chain_h10sz", "h10c"))
    assert len(fx.events) == 114 and len(fx.ref_energy) == 18
    dm = DMRG(fx, "sz")
    dm.init_environments()
    assert dm.n_sites == 10
    e0 = dm.sweep(0, True)
    e1 = dm.sweep(1, False)
    assert fx.pos == len(fx.events)
    worst = max(abs(dm.energies[k] - ref) for k, ref in fx.ref_energy.items())
    print("site energies", ["%.10f" % e for e in e0 + e1], "worst |dE| = %.2e" % worst)
    assert worst < 1e-7
    assert abs(min(e0 + e1) - fx.final_energy) < 1e-7
