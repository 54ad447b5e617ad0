"""Host mirror of the reference's sum-MPO parallel layer for the H·psi path.

* ``ParallelCommunicator`` keeps the names and argument meaning of block2's virtual interface
  (src/core/parallel_rule.hpp:38-308; MPI bodies src/core/parallel_mpi.hpp:125-141, 300-309): ``allreduce_sum``
  (in-place SUM), ``broadcast(owner)``, ``barrier``, ``size`` / ``rank`` / ``root`` / ``is_root``, ``tcomm``.
  The product transport is the C ABI's RCCL communicator (``b2x_comm_*`` / ``b2x_allreduce_sum`` in include/b2x.h)
  acting on device-resident fp64 vectors, one process per GPU.  A second transport over ``torch.distributed``'s gloo
  backend on host arrays exists for the CPU test-suite and for rehearsing N ranks on a single card (RCCL refuses two
  ranks on one device); it is selected explicitly, never silently.  With size == 1 every collective raises, like the
  reference's base class (parallel_rule.hpp:56-307), so serial code can carry a communicator without ever using it.
* ``ParallelRuleSimple`` is the partition rule itself: ``index_prefactor`` of src/dmrg/parallel_simple.hpp:56-99, i.e.
  which share of every one- and two-electron integral a rank keeps when it builds the MPO of ITS Hamiltonian H_r
  (sum over ranks = 1); ``ParallelFCIDUMP.t / .v`` (:104-133) are the masked integrals.
* ``ParallelTensorFunctions`` is the data flow of src/core/parallel_tensor_functions.hpp:51-55: local H_r psi on the
  device, then one in-place all-reduce of sigma.
"""
import time

import numpy as np


class ParallelCommunicator:
    """size / rank / root as in the reference; the transport decides where the vectors live:
    ``from_rccl``: device pointers (the product path); ``from_gloo``: host numpy arrays (tests, rehearsal)."""

    def __init__(self, size=1, rank=0, root=0, rccl=None, gloo_group=None, use_gloo=False):
        self.size, self.rank, self.root = size, rank, root
        self._rccl, self._gloo, self._gloo_group = rccl, use_gloo, gloo_group
        self.tcomm = 0.0  # seconds spent in collectives (Tcomm of the reference)

    @classmethod
    def from_rccl(cls, rank, size, id_file, root=0):
        """RCCL over xGMI through the C ABI (b2x_comm_init: rank 0 publishes the RCCL id in ``id_file``)."""
        from . import capi

        return cls(size, rank, root, rccl=capi.Comm(rank, size, id_file=id_file))

    @classmethod
    def from_gloo(cls, root=0, group=None):
        """torch.distributed (gloo) on HOST arrays: CPU tests / several ranks sharing one card."""
        import torch.distributed as dist

        return cls(dist.get_world_size(group), dist.get_rank(group), root, gloo_group=group, use_gloo=True)

    def is_root(self):
        return self.rank == self.root

    def _need_peers(self):
        if self._rccl is None and not self._gloo:  # the serial base class of the reference: no transport, no collectives
            raise RuntimeError("ParallelCommunicator: collective called on a serial communicator (size == 1)")

    def allreduce_sum(self, data, n=None, stream=0):
        """in-place SUM over ranks.  RCCL transport: ``data`` is a device address of ``n`` doubles (asynchronous, ordered
        on ``stream``); gloo transport: ``data`` is a contiguous float64 numpy array."""
        self._need_peers()
        t = time.perf_counter()
        if self._rccl is not None:
            self._rccl.allreduce_sum(data, n, stream)
        else:
            import torch
            import torch.distributed as dist

            assert isinstance(data, np.ndarray) and data.dtype == np.float64 and data.flags.c_contiguous
            dist.all_reduce(torch.from_numpy(data), op=dist.ReduceOp.SUM, group=self._gloo_group)
        self.tcomm += time.perf_counter() - t
        return data

    def broadcast(self, data, owner, n=None, stream=0):
        self._need_peers()
        t = time.perf_counter()
        if self._rccl is not None:
            self._rccl.broadcast(data, n, owner, stream)
        else:
            import torch
            import torch.distributed as dist

            dist.broadcast(torch.from_numpy(data), src=owner, group=self._gloo_group)
        self.tcomm += time.perf_counter() - t
        return data

    # collectives on DEVICE vectors whatever the transport: RCCL moves them where they are; gloo (ranks that share one card,
    # which RCCL refuses: rehearsals and tests) bounces them through a host array
    def allreduce_device(self, dev_ptr, n, stream=0):
        if self._rccl is not None:
            return self.allreduce_sum(dev_ptr, n, stream)
        from . import capi

        host = np.empty(int(n))
        capi.check(capi.lib().b2x_memcpy_d2h(capi._ptr(host), capi.C.c_void_p(int(dev_ptr)), capi.C.c_size_t(int(n) * 8)))
        self.allreduce_sum(host)
        capi.check(capi.lib().b2x_memcpy_h2d(capi.C.c_void_p(int(dev_ptr)), capi._ptr(host), capi.C.c_size_t(int(n) * 8)))

    def broadcast_device(self, dev_ptr, n, owner, stream=0):
        if self._rccl is not None:
            return self.broadcast(dev_ptr, owner, n, stream)
        from . import capi

        host = np.empty(int(n))
        if self.rank == owner:
            capi.check(capi.lib().b2x_memcpy_d2h(capi._ptr(host), capi.C.c_void_p(int(dev_ptr)), capi.C.c_size_t(int(n) * 8)))
        self.broadcast(host, owner)
        if self.rank != owner:
            capi.check(capi.lib().b2x_memcpy_h2d(capi.C.c_void_p(int(dev_ptr)), capi._ptr(host), capi.C.c_size_t(int(n) * 8)))

    def barrier(self):
        self._need_peers()
        if self._rccl is not None:
            self._rccl.barrier()
        else:
            import torch.distributed as dist

            dist.barrier(group=self._gloo_group)

    def close(self):
        if self._rccl is not None:
            self._rccl.close()
            self._rccl = None


class ParallelRuleSimple:
    """ParallelRuleSimple<S, FL>::index_prefactor (src/dmrg/parallel_simple.hpp:56-99).  Modes as
    ParallelSimpleTypes: "I", "J", "IJ", "KL", "None"."""

    def __init__(self, mode, comm):
        assert mode in ("I", "J", "IJ", "KL", "None")
        self.mode, self.comm = mode, comm

    def _mine(self, x):
        return 1.0 if self.comm.rank == x % self.comm.size else 0.0

    def index_prefactor(self, i, j, k=None, l=None):
        m = self.mode
        if k is None:  # one-electron integral t(i, j)
            if m == "I":
                return self._mine(i)
            if m == "J":
                return self._mine(j)
            if m in ("IJ", "KL"):
                return 0.5 * (self._mine(i) + self._mine(j))
            return 1.0
        ii, jj, kk, ll = sorted((i, j, k, l))  # two-electron integral v(i, j, k, l)
        if m == "I":
            return self._mine(i)
        if m == "J":
            return self._mine(j)
        if m == "IJ":
            return self._mine(jj) if jj == kk else self._mine(jj * (jj + 1) // 2 + ii)
        if m == "KL":
            return self._mine(kk) if jj == kk else self._mine(ll * (ll + 1) // 2 + kk)
        return 1.0

    def is_root(self):
        return self.comm.is_root()


class ParallelFCIDUMP:
    """ParallelFCIDUMP<S, FL> (src/dmrg/parallel_simple.hpp:104-133): the integrals of H_r.  ``t`` is an (n, n) array,
    ``v`` an (n, n, n, n) array in the FCIDUMP's (ij|kl) order; the constant stays with the root
    (pyblock2/driver/core.py:1381-1384)."""

    def __init__(self, t, v, const_e, rule):
        n = t.shape[0]
        self.t = np.array([[rule.index_prefactor(i, j) * t[i, j] for j in range(n)] for i in range(n)])
        pre = np.array([[[[rule.index_prefactor(i, j, k, l) for l in range(n)] for k in range(n)] for j in range(n)]
                        for i in range(n)])
        self.v = pre * v
        self.const_e = const_e if rule.is_root() else 0.0


class ParallelTensorFunctions:
    """ParallelTensorFunctions::operator() (src/core/parallel_tensor_functions.hpp:51-55): sigma = sum_r H_r psi.
    ``plan`` is this rank's capi.Plan (the plan block2 records from the rank's own MPO and environments)."""

    def __init__(self, plan, comm):
        self.plan, self.comm = plan, comm

    def __call__(self, psi_ptr, sigma_ptr, scale=1.0, stream=0):
        self.plan.execute_device(psi_ptr, sigma_ptr, scale, stream)
        if self.comm.size > 1:
            self.comm.allreduce_sum(sigma_ptr, self.plan.sigma_len, stream)
