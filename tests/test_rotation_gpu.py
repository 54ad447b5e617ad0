"""Environment rotation on the MI355X path (through the C ABI / the C++ host mirror) vs the rotated operators the real
reference computed, and vs the oracle on the Cr2 M=250 rotation structure.  fp64, 1e-12 relative to max|result|."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, fill_plan
from block2_preview_amd import synth
from block2_preview_amd.planfile import read_arrays, read_struct_npz
from oracle import oracle
from test_rotation import EROT, _sym

pytestmark = pytest.mark.gpu
ROTSTRUCT = sorted(glob.glob(os.path.join(GOLDEN, "*.rotstruct.npz")))


@pytest.mark.parametrize("fn", EROT, ids=os.path.basename)
def test_symbolic_rotate_on_device(gpu, fn):
    """TensorFunctions::left_rotate / right_rotate of the host mirror, executed by BatchGEMMSeq::rotate_perform"""
    from block2_preview_amd import b2x_host

    d = read_arrays(fn)
    _, v = b2x_host.symbolic_rotate(_sym(fn), d, True)
    assert np.abs(v - d["v_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["v_ref"]).max())


@pytest.mark.parametrize("fn,f", [(fn, f) for fn in ROTSTRUCT for f in (1, 2)],
                         ids=lambda x: os.path.basename(x) if isinstance(x, str) else "x%d" % x)
def test_cr2_rotation_structure(gpu, fn, f):
    """Cr2/SVP M=250 rotation (pair structure recorded by the reference), dimensions x f, synthetic data: vs oracle"""
    pf = fill_plan(synth.scale_plan(read_struct_npz(fn), f), 11)
    ref = np.zeros(pf.sigma_len)
    oracle.replay(pf.pairs, pf.arena, pf.psi, ref, 1.0, 8)
    arena = gpu.Arena.from_host([pf.arena])
    plan = gpu.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len)
    out = np.zeros(pf.sigma_len)
    plan.execute_host(pf.psi, out, 1.0)
    st = plan.stats
    plan.close(), arena.close()
    assert st["macs"] == pf.macs
    assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def _chain_pairs():
    out = []
    for blk in sorted(glob.glob(os.path.join(GOLDEN, "blk_*.blk"))):
        rot = blk.replace("blk_", "rot_", 1).replace("blk.blk", "rot.plan")
        if os.path.exists(rot):
            out.append((blk, rot))
    return out


@pytest.mark.parametrize("blk,rot", _chain_pairs(), ids=lambda x: os.path.basename(x))
def test_left_contract_rotate_device_resident(gpu, blk, rot):
    """MovingEnvironment::left_contract_rotate / right_contract_rotate (src/dmrg/moving_environment.hpp:226-420) as one
    device-resident chain: blocking writes the enlarged operators into an HBM vector, the rotation plan reads that very
    vector as its psi and writes the rotated operators to HBM; only the final result is downloaded and compared with
    the rotated operators the reference computed (both fixtures come from ONE reference run, so the reference's
    enlarged operators are exactly the rotation's input)."""
    from block2_preview_amd.planfile import OUTER_TERM_DTYPE, read_plan

    d = read_arrays(blk)
    terms = np.frombuffer(d["terms"].tobytes(), OUTER_TERM_DTYPE)
    pf = read_plan(rot)
    assert pf.psi_len == int(d["lens"][3])
    site_ops = gpu.Arena.from_host([d["arena"]])
    block_ops = gpu.DeviceBuffer(len(d["in"]), d["in"])
    enlarged = gpu.DeviceBuffer(pf.psi_len)
    gpu.outer_build(site_ops, terms, block_ops.ptr, enlarged.ptr, True, len(d["in"]), pf.psi_len)
    mps = gpu.Arena.from_host([pf.arena])
    plan = gpu.Plan(mps, pf.pairs, pf.psi_len, pf.sigma_len)
    rotated = gpu.DeviceBuffer(pf.sigma_len)
    plan.execute_device(enlarged.ptr, rotated.ptr, 1.0)
    gpu.device_sync()
    out = rotated.download()
    assert np.abs(out - pf.sigma_ref).max() <= 1e-12 * max(1.0, np.abs(pf.sigma_ref).max())
    # and the intermediate really is the reference's enlarged block
    assert np.abs(enlarged.download() - d["out_ref"]).max() <= 1e-12 * max(1.0, np.abs(d["out_ref"]).max())
    for x in (plan, mps, site_ops, block_ops, enlarged, rotated):
        x.close()
