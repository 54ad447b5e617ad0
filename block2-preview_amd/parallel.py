"""Host mirror of the reference's communicator for the sum-MPO path.

``ParallelCommunicator`` keeps the names and argument meaning of block2's virtual interface
(src/core/parallel_rule.hpp:38-308; MPI body src/core/parallel_mpi.hpp:125-141, 300-309):
``allreduce_sum`` (in-place SUM), ``broadcast(root)``, ``barrier``, ``size``/``rank``/``root``.
Transport is torch.distributed: backend "nccl" (= RCCL over xGMI) for device tensors, "gloo" on CPU.
One process per GPU.  With size == 1 every collective raises, like the reference's base class
(parallel_rule.hpp:56-307), so serial code can carry a communicator without ever using it.
"""
import numpy as np

from . import synth


class ParallelCommunicator:
    def __init__(self, size=1, rank=0, root=0, group=None):
        self.size, self.rank, self.root, self.group = size, rank, root, group
        self.tcomm = 0.0  # seconds spent in collectives (Tcomm of the reference)

    @classmethod
    def from_torch_distributed(cls, root=0):
        import torch.distributed as dist

        return cls(dist.get_world_size(), dist.get_rank(), root)

    def _need_peers(self):
        if self.size == 1:
            raise RuntimeError("ParallelCommunicator: collective called with size == 1")

    def allreduce_sum(self, tensor):
        """in-place sum over ranks of a torch tensor (device tensor -> RCCL, CPU tensor -> gloo)"""
        import time

        import torch.distributed as dist

        self._need_peers()
        t = time.perf_counter()
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        self.tcomm += time.perf_counter() - t
        return tensor

    def broadcast(self, tensor, owner):
        import torch.distributed as dist

        self._need_peers()
        dist.broadcast(tensor, src=owner, group=self.group)
        return tensor

    def barrier(self):
        import torch.distributed as dist

        self._need_peers()
        dist.barrier(group=self.group)


class ParallelRuleSumMPO:
    """Which operator terms a rank owns (role of ParallelRuleSimple::index_prefactor,
    src/dmrg/parallel_simple.hpp:56-99, at the level this path sees: the plan's left-operator blocks)."""

    def __init__(self, comm):
        self.comm = comm

    def local_pairs(self, pairs):
        return synth.shard_pairs(pairs, self.comm.rank, self.comm.size)

    def is_root(self):
        return self.comm.rank == self.comm.root
