"""Child process of tests/test_sweep_gpu.py::test_sum_mpo_sweep_one_rank_per_process: ONE rank of a sum-MPO DMRG calculation.
It replays ITS rank's event chain of the reference's mpirun run with sweep.DMRG; sigma, the diagonal and the perturbed
wavefunctions are summed over the ranks by the communicator.  With world > 1 the ranks share card 0 on the test box, which RCCL
refuses, so the communicator is the host mirror's gloo transport (device vectors bounced through the host); with world == 1 the
RCCL communicator of the C ABI carries the same calls.
usage: sum_mpo_sweep_worker.py <rank> <world> <port> <chain prefix with %d for the rank> <sym> <n_sweeps> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    prefix, sym, n_sweeps, out = sys.argv[4], sys.argv[5], int(sys.argv[6]), sys.argv[7]
    from block2_preview_amd import capi
    from block2_preview_amd.parallel import ParallelCommunicator
    from block2_preview_amd.sweep import DMRG, ChainFixture

    capi.device_init(0)
    if world > 1:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = ParallelCommunicator.from_gloo()
    else:
        comm = ParallelCommunicator.from_rccl(0, 1, None)
    fx = ChainFixture(prefix % rank)
    dm = DMRG(fx, sym)
    dm.comm = comm
    dm.init_environments()
    for isw in range(n_sweeps):
        dm.sweep(isw, isw % 2 == 0)
    assert fx.pos == len(fx.events)
    res = {"rank": rank, "energies": {"%d,%d" % k: v for k, v in dm.energies.items()},
           "ndav": {"%d,%d" % k: v for k, v in dm.ndav.items()}, "tcomm": comm.tcomm,
           "starts": sorted(set(v[0] for v in dm.guess_log.values()))}
    json.dump(res, open(out + ".r%d" % rank, "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
