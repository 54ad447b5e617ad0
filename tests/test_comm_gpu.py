"""The sum-MPO communicator of the C ABI on the device, and the N > 1 product path with the HIP kernels:
 * b2x_comm_* over RCCL with one rank (all the test box can host: RCCL refuses two ranks on one card): init through the
   id file, in-place all-reduce / broadcast of a device vector, barrier, ordering on the caller's stream;
 * two FRESH child processes, each one sum-MPO rank running ITS plan on the GPU through the C ABI; the partial sigmas
   are summed (gloo: the ranks share card 0) and must equal the all-reduced sigma of the reference — for the plans the
   reference itself recorded on 2 MPI ranks (ParallelRuleSimple IJ) and for a golden plan sharded by operator block."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_communicator_single_rank(gpu, tmp_path):
    comm = gpu.Comm(0, 1, id_file=str(tmp_path / "id"))
    x = np.linspace(-1.0, 1.0, 100003)
    d = gpu.DeviceBuffer(len(x), x)
    comm.allreduce_sum(d.ptr, len(x))  # one rank: the sum is the vector itself
    comm.broadcast(d.ptr, len(x), 0)
    comm.barrier()
    gpu.device_sync()
    assert np.array_equal(d.download(), x)
    # ordering on the caller's stream: scale -> all-reduce -> scale, no explicit synchronisation in between
    import ctypes as C

    gpu.check(gpu.lib().b2x_vec_scal(C.c_double(2.0), C.c_void_p(d.ptr), C.c_size_t(len(x)), None))
    comm.allreduce_sum(d.ptr, len(x))
    gpu.check(gpu.lib().b2x_vec_scal(C.c_double(-0.5), C.c_void_p(d.ptr), C.c_size_t(len(x)), None))
    gpu.device_sync()
    assert np.array_equal(d.download(), -x)
    with pytest.raises(gpu.B2XError):
        gpu.Comm(1, 1, id_file=str(tmp_path / "id2"))  # rank out of range
    # explicit id exchange (a launcher with its own broadcast)
    c2 = gpu.Comm(0, 1, id_bytes=gpu.Comm.unique_id())
    c2.barrier()
    c2.close(), comm.close(), d.close()


def test_davidson_with_communicator(gpu, tmp_path):
    """davidson with pcomm (iterative_matrix_functions.hpp:968-970, 1162-1167) on the device: sigma all-reduced after
    every H.psi, new basis vectors and the result broadcast from root — real RCCL calls on a one-rank communicator —
    gives the eigenpair of the run without a communicator."""
    from block2_preview_amd import b2x_host
    from block2_preview_amd.planfile import read_plan

    pf = read_plan(os.path.join(GOLDEN, "n2su2.sw2.site5.plan"))
    arena = gpu.Arena.from_host([pf.arena])
    plan = gpu.Plan(arena, pf.pairs, pf.psi_len, pf.sigma_len)
    diag = gpu.DeviceBuffer(pf.psi_len, pf.diag)
    comm = gpu.Comm(0, 1, id_file=str(tmp_path / "id"))
    res = []
    for c in (None, (comm._h.value, 0, 1, 0)):
        ket = gpu.DeviceBuffer(pf.psi_len, pf.psi)
        e, nd = b2x_host.davidson_device(plan._h.value, diag.ptr, ket.ptr, pf.psi_len, 1e-12, 500, comm=c)
        res.append((e, nd, ket.download()))
        ket.close()
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
    comm.close(), diag.close(), plan.close(), arena.close()


def _spawn(world, shard, fns, two_stage=0):
    port = str(_free_port())
    env = dict(os.environ, B2X_TEST_TWO_STAGE=str(two_stage))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sum_mpo_worker.py"), str(r), str(world), port,
                               str(shard)] + fns, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(line) == 1, outs[0][0]
    return json.loads(line[0])


@pytest.mark.parametrize("tag,two_stage", [("sw1.site5", 0), ("sw2.site4", 0), ("sw2.site4", 1)])
def test_two_ranks_hip_path_reference_partition(gpu, tag, two_stage):
    fns = [os.path.join(GOLDEN, "n2su2_ij.r%dof2.%s.plan" % (r, tag)) for r in range(2)]
    j = _spawn(2, 0, fns, two_stage)
    assert j["fallback"] == 0
    assert j["err"] <= 1e-12 * max(1.0, j["max"]), j
    assert j["part_err"] > 1e-3 * j["max"]  # a single rank's H_r psi is NOT H psi


@pytest.mark.parametrize("name", ["n2sz.sw2.site4.plan", "h10szm50.sw1.site5.plan"])
def test_two_ranks_hip_path_sharded_plan(gpu, name):
    fn = os.path.join(GOLDEN, name)
    j = _spawn(2, 1, [fn])
    assert 0 < j["n_mine"] < j["n_all"]
    assert j["err"] <= 1e-12 * max(1.0, j["max"]), j


def test_one_rank_rccl_allreduce_of_hpsi(gpu):
    """the product data flow with the RCCL communicator in the loop (world = 1): device-resident H psi + all-reduce"""
    j = _spawn(1, 0, [os.path.join(GOLDEN, "n2su2.sw2.site5.plan")])
    assert j["err"] <= 1e-12 * max(1.0, j["max"]), j
