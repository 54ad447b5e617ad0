import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The full-size GPU tests generate their data on the device with torch.  torch must load and initialise ITS HIP
    # runtime before libb2x.so pulls in the system one (the other order leaves torch without a device), so when the GPU
    # tests are selected torch is imported here, before any fixture touches libb2x.
    expr = config.getoption("markexpr", "") or ""
    if ("gpu" in expr and "not gpu" not in expr) or (os.path.exists("/dev/kfd") and "not gpu" not in expr):
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass


def golden_plan_files():
    """single-process golden plans (sigma_ref = H psi of THIS plan); the per-rank plans of the reference's 2-rank sum-MPO
    run (*.r<k>of<n>.*: sigma_ref is the all-reduced sum) are used by the sum-MPO tests only"""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.plan")) if ".r0of" not in f and ".r1of" not in f)


@pytest.fixture(scope="session")
def built():
    """libb2x.so + liboracle.so present (built in-tree; hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge

    ge.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    from block2_preview_amd import capi

    if capi.device_count() == 0:
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    capi.device_init(0)
    return capi


def fill_plan(pf, seed):
    """uniform [0,1) operator / psi data, like Random::fill of the reference (src/core/utils.hpp:247-252)"""
    rng = np.random.default_rng(seed)
    pf.arena = rng.random(pf.arena_len)
    pf.psi = rng.random(pf.psi_len)
    return pf
