"""Symbolic layer end to end on the MI355X: infos + operator tensors + H_eff terms  ->  initialize_wfn  ->
tensor_product_multiply (records the plan)  ->  device replay  ==  the reference's sigma; and eigs() reaches the
energy the reference's Davidson reported for that site."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from block2_preview_amd.planfile import read_arrays

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN, "e_*.eham")))


@pytest.fixture(scope="module")
def host(gpu):
    from block2_preview_amd import b2x_host

    b2x_host.device_init(0)
    return b2x_host


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_symbolic_hpsi_on_device(host, fn):
    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian("su2" if "su2" in os.path.basename(fn) else "sz", d)
    sig = np.zeros(len(d["sigma_ref"]))
    h(d["psi"].copy(), sig, 1.0)
    assert h.n_pairs == int(d["n_pairs"][0])
    assert np.abs(sig - d["sigma_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["sigma_ref"]).max())
    # sigma += factor * H psi
    sig2 = np.ones(len(sig))
    h(d["psi"].copy(), sig2, -0.5)
    assert np.abs(sig2 - (1.0 - 0.5 * d["sigma_ref"])).max() <= 1e-12 * max(1.0, np.abs(d["sigma_ref"]).max())


def test_symbolic_eigs_site_energy(host):
    fn = [f for f in FILES if "e_n2su2.sw1.site5" in f][0]
    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian("su2", d)
    e, ndav, nflop, tdav, ket = h.eigs(d["psi"].tolist(), 1e-12, 500)
    # N2/STO-3G: the sweep-1 site energy of the generating run is the converged ground state
    assert abs(e + float(d["const_e"][0]) - (-107.654122447525)) < 1e-6
    assert nflop == ndav * int(d["n_pairs"][1])


@pytest.mark.parametrize("fn", FILES, ids=[os.path.basename(f) for f in FILES])
def test_diag_build_on_device(host, fn):
    """diag of H_eff built on the device (b2x_diag_build) == the reference's diag (EffectiveHamiltonian ctor)"""
    d = read_arrays(fn)
    h = host.SymbolicEffectiveHamiltonian("su2" if "su2" in os.path.basename(fn) else "sz", d)
    diag = np.asarray(h.compute_diag())
    assert np.abs(diag - d["diag"]).max() <= 1e-12 * max(1.0, np.abs(d["diag"]).max())
    again = np.asarray(h.compute_diag())
    assert np.array_equal(diag, again)  # deterministic
