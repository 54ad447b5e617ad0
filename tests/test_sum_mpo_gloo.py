"""N > 1 path on the CPU: two gloo ranks shard the operator terms of a golden plan (sum-MPO), each
replays ITS share with the oracle, and one all-reduce of the partial sigma reproduces the reference's
full sigma — the data flow of ParallelTensorFunctions::operator() (parallel_tensor_functions.hpp:51-55)
and of unit_test/mpi/test_sum_mpo_n2_sto3g.cpp (2 local ranks)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from block2_preview_amd import synth
    from block2_preview_amd.parallel import ParallelCommunicator, ParallelRuleSumMPO
    from block2_preview_amd.planfile import read_plan
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = ParallelCommunicator.from_torch_distributed()
    rule = ParallelRuleSumMPO(comm)
    pf = read_plan(fn)
    mine = rule.local_pairs(pf.pairs)
    sig = np.zeros(pf.sigma_len)
    oracle.replay(mine, pf.arena, pf.psi, sig)
    t = torch.from_numpy(sig)
    comm.allreduce_sum(t)
    comm.barrier()
    if rule.is_root():
        q.put((len(mine), len(pf.pairs), float(np.abs(t.numpy() - pf.sigma_ref).max()), float(np.abs(pf.sigma_ref).max())))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["n2sz.sw2.site4.plan", "h10szm50.sw1.site5.plan"])
def test_two_rank_sum_mpo_allreduce(name):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    fn = os.path.join(GOLDEN, name)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fn, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    n_mine, n_all, err, mx = q.get(timeout=10)
    assert 0 < n_mine < n_all  # the terms really were split
    assert err <= 1e-12 * max(1.0, mx)


def test_single_rank_communicator_refuses_collectives():
    from block2_preview_amd.parallel import ParallelCommunicator

    with pytest.raises(RuntimeError):
        ParallelCommunicator().barrier()
