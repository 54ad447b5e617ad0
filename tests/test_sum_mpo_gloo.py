"""N > 1 path on the CPU: two gloo ranks, each replays ITS plan with the oracle, and one all-reduce of the partial
sigma through the host mirror's ParallelCommunicator reproduces the reference's all-reduced sigma — the data flow of
ParallelTensorFunctions::operator() (parallel_tensor_functions.hpp:51-55).  Two kinds of per-rank plans:
 * the plans the REFERENCE ITSELF recorded on 2 MPI ranks with ParallelRuleSimple(IJ) (tests/golden/n2su2_ij.r*of2.*.plan:
   every rank's own MPO, environments and plan; sigma_ref = the all-reduced H psi the reference computed) — this pins the
   partition; and
 * one golden plan sharded by left-operator block (what bench.py does with a synthetic plan).
The partition rule itself (index_prefactor) is checked against the table the reference printed for both ranks."""
import glob
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


def _worker(rank, world, port, fns, shard, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from block2_preview_amd import synth
    from block2_preview_amd.parallel import ParallelCommunicator
    from block2_preview_amd.planfile import read_plan
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = ParallelCommunicator.from_gloo()
    pf = read_plan(fns[rank])
    mine = synth.shard_pairs(pf.pairs, rank, world) if shard else pf.pairs
    sig = np.zeros(pf.sigma_len)
    oracle.replay(mine, pf.arena, pf.psi, sig)
    part = sig.copy()
    comm.allreduce_sum(sig)
    comm.barrier()
    if comm.is_root():
        q.put((len(mine), len(pf.pairs), float(np.abs(sig - pf.sigma_ref).max()), float(np.abs(pf.sigma_ref).max()),
               float(np.abs(part - pf.sigma_ref).max())))
    dist.destroy_process_group()


def _run(fns, shard):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fns, shard, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return q.get(timeout=10)


@pytest.mark.parametrize("tag", ["sw1.site5", "sw2.site4"])
def test_reference_two_rank_partition(tag):
    fns = [os.path.join(GOLDEN, "n2su2_ij.r%dof2.%s.plan" % (r, tag)) for r in range(2)]
    n_mine, n_all, err, mx, part_err = _run(fns, False)
    assert err <= 1e-12 * max(1.0, mx)
    assert part_err > 1e-3 * mx  # one rank alone does NOT give H psi: the sum is needed


@pytest.mark.parametrize("name", ["n2sz.sw2.site4.plan", "h10szm50.sw1.site5.plan"])
def test_two_rank_sharded_plan_allreduce(name):
    fn = os.path.join(GOLDEN, name)
    n_mine, n_all, err, mx, _ = _run([fn, fn], True)
    assert 0 < n_mine < n_all  # the terms really were split
    assert err <= 1e-12 * max(1.0, mx)


def test_single_rank_communicator_refuses_collectives():
    from block2_preview_amd.parallel import ParallelCommunicator

    with pytest.raises(RuntimeError):
        ParallelCommunicator().barrier()


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(GOLDEN, "*.prefactors"))), ids=os.path.basename)
def test_index_prefactor_matches_reference(fn):
    """ParallelRuleSimple.index_prefactor == the table ParallelRuleSimple<S,FL>::index_prefactor produced inside the
    reference's 2-rank run, for every (i, j) and (i, j, k, l); the ranks' shares sum to one"""
    from block2_preview_amd.parallel import ParallelCommunicator, ParallelRuleSimple
    from block2_preview_amd.planfile import read_arrays

    d = read_arrays(fn)
    n, rank, size, mode = (int(x) for x in d["meta"])
    rule = ParallelRuleSimple({1: "I", 3: "IJ"}[mode], ParallelCommunicator(size, rank))
    ij = np.array([[rule.index_prefactor(i, j) for j in range(n)] for i in range(n)])
    assert np.array_equal(ij.ravel(), d["ij"])
    ijkl = np.array([rule.index_prefactor(i, j, k, l) for i in range(n) for j in range(n) for k in range(n) for l in range(n)])
    assert np.array_equal(ijkl, d["ijkl"])
    tot = sum(np.array([ParallelRuleSimple({1: "I", 3: "IJ"}[mode], ParallelCommunicator(size, r)).index_prefactor(i, j, k, l)
                        for i in range(n) for j in range(n) for k in range(0, n, 3) for l in range(0, n, 2)]) for r in range(size))
    assert np.array_equal(tot, np.ones_like(tot))
