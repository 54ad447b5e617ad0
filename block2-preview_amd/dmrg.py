"""block2's sweep layer under block2's names, over the device-resident chain engine of sweep.py.

    MovingEnvironment(mpo, bra, ket, tag)      src/dmrg/moving_environment.hpp:59-, pybind src/pybind/pybind_dmrg.hpp:773-900
        .init_environments()  .move_to(i)  .eff_ham(...)  .left_contract_rotate(i)  .right_contract_rotate(i)
    EffectiveHamiltonian (what eff_ham returns)   .eigs(...)  .perturbative_noise(...)  .deallocate()
                                               src/dmrg/effective_hamiltonian.hpp:124-558, pybind :588-640
    DMRG(me, bond_dims, noises)                src/dmrg/sweep_algorithm.hpp:60-3230, pybind :1298-1380
        .update_two_dot(i, forward, bond_dim, noise, davidson_conv_thrd)  .blocking(...)  .sweep(...)  .solve(n_sweeps, forward, tol)
        + the tuning fields (davidson_conv_thrds, davidson_max_iter, noise_type, trunc_type, decomp_type, cutoff, iprint)
          and the records (energies, discarded_weights, sweep_energies, sweep_discarded_weights, sweep_time, ...)
    DMRGDriver.dmrg(mpo, ket, n_sweeps, tol, bond_dims, noises, thrds, iprint, cutoff, dav_max_iter)
                                               src/dmrg/dmrg_driver.hpp:415-464 (C++), pyblock2/driver/core.py:4600-4760

What stands where block2 has its MPO / MPS objects: the SYMBOLIC side of a calculation (operator infos, expressions of the
enlarged operators and of H_eff, quantum-number bookkeeping of the bonds) belongs to block2's MPO / Partition / MPSInfo
layers, which are out of this build's scope; it comes from the event chain one reference run of the same system
recorded (sweep.ChainFixture).  `MPO` and `MPS` below are the handles to it: MPO = the chain (site operators, expressions,
constant), MPS = the state being optimised (tensors produced by this code's own splits; the starting tensors from the chain).
Everything NUMERIC — every operator block of every environment, H.psi, Davidson, noise, density matrices, the new MPS
tensors, the energies — is computed here, on the device, as sweep.py describes.  Consequences, checked and reported instead
of silently assumed: the schedule (bond dimensions, which sweeps carry noise, the direction of the first sweep) must be the one
the chain was recorded with — a different one raises; the per-sector bond dimensions after each split are the chain's.
"""
import time

import numpy as np

from . import capi
from .sweep import DMRG as _Engine, ChainFixture


# ---- enums (the subset this path implements; the values are block2's, src/core/threading.hpp / src/dmrg/*.hpp) ------------
class FuseTypes:
    NoFuseL, NoFuseR, FuseL, FuseR, FuseLR = 4, 8, 1, 2, 3


class NoiseTypes:
    Zero, Wavefunction, DensityMatrix, Perturbative = 0, 1, 2, 4
    Collected, Reduced, Unscaled, LowMem, MidMem = 8, 16, 32, 64, 128
    ReducedPerturbative = 4 | 16


class TruncationTypes:
    Physical, Reduced, ReducedInversed, KeepOne = 0, 1, 2, 4
    RealDensityMatrix = 1 << 30


class DecompositionTypes:
    DensityMatrix, SVD, PureSVD = 0, 1, 2


class MPO:
    """handle to the symbolic side of one Hamiltonian: the event chain of a reference run (see the module docstring)"""

    def __init__(self, chain_prefix, sym):
        self.fixture = chain_prefix if isinstance(chain_prefix, ChainFixture) else ChainFixture(chain_prefix)
        self.sym, self.const_e, self.n_sites = sym, None, None

    @property
    def tag(self):
        return "HQC"


class ParallelMPO:
    """block2's ParallelMPO (src/dmrg/parallel_mpo.hpp): the MPO of a sum-MPO calculation, H = sum_r H_r with the integrals
    partitioned by a ParallelRule (ParallelRuleSimple: src/dmrg/parallel_simple.hpp:56-99).  Here: the chains of ALL ranks of
    one reference run under mpirun (one MPO handle per rank) + the rule.  A MovingEnvironment over it carries every rank's
    environments and solves each site over the sum of the ranks' plans — across GPUs that sum is the all-reduce of sigma
    (ParallelTensorFunctions::operator(), src/core/parallel_tensor_functions.hpp:51-55)."""

    def __init__(self, rank_mpos, rule=None):
        self.ranks, self.rule = list(rank_mpos), rule
        self.sym, self.fixture, self.const_e, self.n_sites = self.ranks[0].sym, self.ranks[0].fixture, None, None

    @property
    def tag(self):
        return "HQC"


class MPSInfo:
    def __init__(self, bond_dim=0, tag="KET"):
        self.bond_dim, self.tag = bond_dim, tag


class MPS:
    """the state being optimised: `tensors[i]` = MPS tensor of site i produced by the last split that touched it (host arrays in
    the layout of the chain's rotation event), `center`, `dot`, `canonical_form` as block2 keeps them"""

    def __init__(self, n_sites=0, center=0, dot=2, info=None):
        self.n_sites, self.center, self.dot = n_sites, center, dot
        self.tensors, self.info = {}, info or MPSInfo()
        self.canonical_form = ""


class Iteration:
    """DMRG::Iteration (sweep_algorithm.hpp:140-170)"""

    def __init__(self, energies, error, mmps, ndav, nflop=0, tdav=0.0):
        self.energies, self.error, self.mmps, self.ndav, self.nflop, self.tdav = energies, error, mmps, ndav, nflop, tdav
        self.quanta = []

    def __repr__(self):
        return "Mmps = %4d Ndav = %3d E = %17.10f Error = %8.2e FLOPS = %8.2e Tdav = %.2f" % (
            self.mmps, self.ndav, self.energies[0], self.error, self.nflop / max(self.tdav, 1e-30), self.tdav)


class EffectiveHamiltonian:
    """what MovingEnvironment.eff_ham returns: the plan of H_eff on the site's operator arena + its diagonal, in HBM"""

    def __init__(self, me, parts, event, noise_event):
        self.me, self._parts, self._event, self._noise_event = me, parts, event, noise_event
        self.ket_len, self.n_pairs = parts[0]["n"], sum(q["n_pairs"] for q in parts)
        self._solved = None

    def eigs(self, iprint=False, conv_thrd=5e-6, max_iter=5000, soft_max_iter=-1, *unused, **kw):
        """-> (energy, ndav, nflop, tdav) as EffectiveHamiltonian::eigs (effective_hamiltonian.hpp:470-558); the constant of
        the Hamiltonian is included in the energy like in DMRG::update_two_dot's report"""
        eng = self.me._eng
        t0 = time.perf_counter()
        eng.conv_thrd = conv_thrd
        for other in self.me._engs[1:]:  # (sum-MPO: the other ranks' next site is prepared during this solve as well)
            other._prefetch_next()
        e, ndav, psi, kinfo, n_pairs = eng._solve(self._parts, self._noise_event)
        tdav = time.perf_counter() - t0
        self._solved = (e, ndav, psi, kinfo)
        self._parts = None  # (the engine released the plans, the arenas and the diagonals)
        nflop = 2 * ndav * self.me._last_macs
        return e, ndav, nflop, tdav

    def perturbative_noise(self, *a, **kw):
        """the perturbed wavefunctions of the site (EffectiveHamiltonian::perturbative_noise): computed on the device right
        after the eigensolver, while the operator arena is resident (see sweep.DMRG._perturb); returns their host copy"""
        return self.me._eng.pket

    def deallocate(self):
        for q in self._parts or []:
            q["plan"].close(), q["arena"].close(), q["arena_t"].close(), q["diag"].close()
        self._parts = None


class MovingEnvironment:
    def __init__(self, mpo, bra, ket, tag="DMRG"):
        if bra is not ket:
            raise NotImplementedError("bra != ket (transition / non-Hermitian environments) is out of this path's scope")
        self.mpo, self.bra, self.ket, self.tag = mpo, bra, ket, tag
        self.dot, self.center, self.n_sites = 2, ket.center, None
        self.delayed_contraction, self.cached_contraction, self.save_partition_info = None, True, False
        self.para_rule, self.iprint = None, 0
        ranks = mpo.ranks if isinstance(mpo, ParallelMPO) else [mpo]
        self._engs = [_Engine(r.fixture, r.sym) for r in ranks]  # one set of environments per sum-MPO rank
        self._eng = self._engs[0]
        self._forward = True
        self._isweep = 0
        self._last_macs = 0

    # block2's timers (moving_environment.hpp:86-88), read from the engine's clock
    tctr = property(lambda self: self._eng.tm.get("block", 0.0))
    trot = property(lambda self: self._eng.tm.get("rotate", 0.0))
    tint = property(lambda self: self._eng.tm.get("transform", 0.0))
    tdiag = property(lambda self: self._eng.tm.get("eff_ham.device", 0.0))

    def init_environments(self, iprint=False):
        """all right blocks of the starting state, last site inwards (moving_environment.hpp:1245-)"""
        blocks = [e.init_environments() for e in self._engs][0]
        assert all(e.n_sites == self._eng.n_sites for e in self._engs)
        self.n_sites = self.ket.n_sites = self.mpo.n_sites = self._eng.n_sites
        self.center = self.ket.center = 0
        return blocks

    def move_to(self, i, preserve_data=False):
        """MovingEnvironment::move_to (moving_environment.hpp:1534-1570): the environments of centre i — the enlarged block
        left behind is rotated with the MPS tensor of the last split, the blocks of the new site are contracted"""
        if self.n_sites is None:
            raise RuntimeError("init_environments() first")
        if i != self.center:
            self._forward = i > self.center
        for e in self._engs:
            e._move_to(i, self._forward)
        self.center = self.ket.center = i

    def left_contract_rotate(self, i):
        """one step to the right: enlarged left block of site i - 1 rotated, block of site i contracted (:226-442)"""
        self._forward = True
        for e in self._engs:
            e._move_to(i, True)
        self.center = self.ket.center = i

    def right_contract_rotate(self, i):
        self._forward = False
        for e in self._engs:
            e._move_to(i, False)
        self.center = self.ket.center = i

    def eff_ham(self, fuse_type=FuseTypes.FuseLR, forward=True, compute_diag=True, bra_wfn=None, ket_wfn=None):
        """H_eff of the current centre (moving_environment.hpp:2062-2201): plan + diagonal on the device"""
        if fuse_type != FuseTypes.FuseLR:
            raise NotImplementedError("only the two-site effective Hamiltonian (FuseTypes.FuseLR) is on this path")
        parts, d0, dn = [], None, []
        for eng in self._engs:
            d = eng._eham_event(self._isweep, self.center)
            if eng.fx.peek() == "enoise":  # (sum-MPO: one perturbative-noise step per rank, summed on the root)
                dn.append(eng.fx.next("enoise")[1])
            parts.append(eng._eff_ham(d))
            d0 = d0 if d0 is not None else d
        self.mpo.const_e = parts[0]["const_e"]
        self._last_macs = sum(int(q["plan"].stats["macs"]) for q in parts)
        return EffectiveHamiltonian(self, parts, d0, (dn if len(dn) > 1 else dn[0]) if dn else None)


class DMRG:
    def __init__(self, me, bond_dims, noises):
        self.me, self.bond_dims, self.noises = me, list(bond_dims), list(noises)
        self.iprint, self.cutoff, self.quanta_cutoff = 2, 1e-14, 1e-3
        self.davidson_conv_thrds, self.davidson_rel_conv_thrd = [], 0.0
        self.davidson_max_iter, self.davidson_soft_max_iter = 5000, -1
        self.davidson_def_min_size, self.davidson_def_max_size = 2, 50
        self.noise_type = NoiseTypes.DensityMatrix
        self.trunc_type = TruncationTypes.Physical
        self.decomp_type = DecompositionTypes.DensityMatrix
        self.energies, self.discarded_weights, self.mps_quanta = [], [], []
        self.sweep_energies, self.sweep_discarded_weights, self.sweep_quanta, self.sweep_time = [], [], [], []
        self.sweep_cumulative_nflop = 0
        self.forward, self.isweep = True, 0
        self.teff = self.teig = self.tprt = self.tblk = self.tmve = self.tdm = self.tsplt = 0.0

    def _check_schedule(self, noise):
        if self.decomp_type != DecompositionTypes.DensityMatrix or (self.trunc_type & ~TruncationTypes.RealDensityMatrix):
            raise NotImplementedError("this path splits with DecompositionTypes.DensityMatrix / TruncationTypes.Physical")
        if noise != 0 and not (self.noise_type & NoiseTypes.Perturbative):
            raise NotImplementedError("noise on this path is the perturbative noise (NoiseTypes.ReducedPerturbative): the "
                                      "random-noise types are not reproducible against a reference run")

    def update_two_dot(self, i, forward, bond_dim, noise, davidson_conv_thrd):
        """DMRG::update_two_dot (sweep_algorithm.hpp:811-1261): effective Hamiltonian, Davidson, [perturbative noise,]
        density-matrix split of the two-site wavefunction at sites i, i + 1"""
        self._check_schedule(noise)
        me, eng = self.me, self.me._eng
        me._isweep = self.isweep
        t0 = time.perf_counter()
        h_eff = me.eff_ham(FuseTypes.FuseLR, forward, True)
        if (h_eff._noise_event is not None) != (noise != 0):
            raise RuntimeError("sweep %d site %d: noise = %g here, but the chain was recorded %s noise at this site" % (
                self.isweep, i, noise, "with" if h_eff._noise_event is not None else "without"))
        self.teff += time.perf_counter() - t0
        t0 = time.perf_counter()
        e, ndav, nflop, tdav = h_eff.eigs(self.iprint >= 3, davidson_conv_thrd, self.davidson_max_iter,
                                          self.davidson_soft_max_iter)
        self.teig += time.perf_counter() - t0
        for g in me._engs:
            g._finish_site(self.isweep, i, e, ndav, h_eff._solved[2], h_eff._solved[3])
            g.pket = eng.pket  # (the perturbed wavefunctions summed over the ranks: every rank splits with the same density matrix)
        h_eff.deallocate()
        t0 = time.perf_counter()
        sp = [g.split_site(forward) for g in me._engs][0]
        self.tsplt += time.perf_counter() - t0
        error, mmps = (sp["error"], sp["mmps"]) if sp else (0.0, 0)
        if mmps > bond_dim:
            raise RuntimeError("the chain keeps %d states at this bond, the schedule allows %d" % (mmps, bond_dim))
        if eng.ahead is not None:
            self.me.ket.tensors[i if forward else i + 1] = eng.ahead[1]
        self.me.ket.info.bond_dim = max(self.me.ket.info.bond_dim, mmps)
        return Iteration([e], error, mmps, ndav, nflop, tdav)

    def blocking(self, i, forward, bond_dim, noise, davidson_conv_thrd):
        """DMRG::blocking (sweep_algorithm.hpp:2473-2548): move the environments to site i, then update the site"""
        t0 = time.perf_counter()
        self.me._forward = forward
        self.me.move_to(i)
        self.tmve += time.perf_counter() - t0
        return self.update_two_dot(i, forward, bond_dim, noise, davidson_conv_thrd)

    def sweep(self, forward, bond_dim, noise, davidson_conv_thrd):
        """DMRG::sweep (sweep_algorithm.hpp:2550-2699) -> (energies of the best site, largest discarded weight, quanta)"""
        n = self.me.n_sites
        self.teff = self.teig = self.tprt = self.tblk = self.tmve = self.tdm = self.tsplt = 0.0
        self.sweep_energies, self.sweep_discarded_weights, self.sweep_quanta = [], [], []
        self.sweep_cumulative_nflop = 0
        self.forward = forward
        sites = range(0, n - 1) if forward else range(n - 2, -1, -1)
        t_sweep = time.perf_counter()
        for i in sites:
            t0 = time.perf_counter()
            r = self.blocking(i, forward, bond_dim, noise, davidson_conv_thrd)
            self.tblk += time.perf_counter() - t0
            self.sweep_cumulative_nflop += r.nflop
            self.sweep_energies.append(r.energies)
            self.sweep_discarded_weights.append(r.error)
            self.sweep_quanta.append(r.quanta)
            if self.iprint >= 2:
                print(" %s Site = %4d-%4d .. %r T = %.2f" % ("-->" if forward else "<--", i, i + 1, r,
                                                               time.perf_counter() - t0), flush=True)
        capi.device_sync()
        self.sweep_time.append(time.perf_counter() - t_sweep)
        idx = int(np.argmin([e[0] for e in self.sweep_energies]))
        return self.sweep_energies[idx], max(self.sweep_discarded_weights), self.sweep_quanta[idx]

    def solve(self, n_sweeps, forward=True, tol=1e-6, sweep_start=0):
        """DMRG::solve (sweep_algorithm.hpp:3032-3230): the sweep schedule; converged when |dE| < tol with the noise and the
        bond dimension at their final values"""
        if len(self.bond_dims) < n_sweeps:
            self.bond_dims += [self.bond_dims[-1]] * (n_sweeps - len(self.bond_dims))
        if len(self.noises) < n_sweeps:
            self.noises += [self.noises[-1] if self.noises else 0.0] * (n_sweeps - len(self.noises))
        for k in range(len(self.davidson_conv_thrds), len(self.noises)):
            self.davidson_conv_thrds.append((self.noises[k] if self.noises[k] != 0 else (tol if tol != 0 else 1e-9)) * 0.1)
        t_start = time.perf_counter()
        for iw in range(sweep_start, n_sweeps):
            self.isweep = iw
            if self.iprint >= 1:
                print("Sweep = %4d | Direction = %8s | Bond dimension = %4d | Noise = %9.2e | Dav threshold = %9.2e" % (
                    iw, "forward" if forward else "backward", self.bond_dims[iw], self.noises[iw],
                    self.davidson_conv_thrds[iw]), flush=True)
            es, dw, qs = self.sweep(forward, self.bond_dims[iw], self.noises[iw], self.davidson_conv_thrds[iw])
            self.energies.append(es), self.discarded_weights.append(dw), self.mps_quanta.append(qs)
            de = self.energies[-1][-1] - self.energies[-2][-1] if len(self.energies) >= 2 else None
            converged = (de is not None and tol > 0 and abs(de) < tol and self.noises[iw] == self.noises[-1]
                         and self.bond_dims[iw] == self.bond_dims[-1])
            forward = not forward
            if self.iprint >= 1:
                print("Time elapsed = %10.3f | E = %18.10f%s | DW = %9.5e" % (
                    time.perf_counter() - t_start, es[0], "" if de is None else " | DE = %6.2e" % de, dw), flush=True)
                if self.iprint >= 2:
                    print("Time sweep = %12.3f | %.3g FLOP/SWP\n | Teff = %.3f | Teig = %.3f | Tblk = %.3f | Tmve = %.3f"
                          " | Tsplt = %.3f" % (self.sweep_time[-1], self.sweep_cumulative_nflop, self.teff, self.teig,
                                               self.tblk, self.tmve, self.tsplt), flush=True)
            if converged or self.me._eng.fx.pos >= len(self.me._eng.fx.events):
                break  # (converged, or the recorded chain ends here)
        self.forward = forward
        return self.energies[-1][0]


class DMRGDriver:
    """the driver entry (C++ DMRGDriver::dmrg, src/dmrg/dmrg_driver.hpp:415-464; Python pyblock2/driver/core.py DMRGDriver.dmrg):
    same arguments, same defaults, same construction of MovingEnvironment / DMRG around them"""

    def __init__(self, scratch="./nodex", symm_type="su2", n_threads=None, stack_mem=None, device=0):
        self.scratch, self.symm_type = scratch, str(symm_type).lower()
        capi.device_init(device)

    def get_chain_mpo(self, chain_prefix):
        return MPO(chain_prefix, self.symm_type)

    def get_chain_mps(self, mpo, tag="KET", bond_dim=0):
        return MPS(center=0, dot=2, info=MPSInfo(bond_dim, tag))

    def dmrg(self, mpo, ket, n_sweeps=10, tol=1e-8, bond_dims=None, noises=None, thrds=None, iprint=0, cutoff=1e-20,
             dav_max_iter=4000):
        bond_dims = list(bond_dims) if bond_dims else [ket.info.bond_dim]
        noises = list(noises) if noises is not None and len(noises) else [1e-5] * 5 + [0.0]
        thrds = list(thrds) if thrds is not None and len(thrds) else [1e-6] * 4 + [1e-7]
        bra = ket
        me = MovingEnvironment(mpo, bra, ket, "DMRG")
        me.cached_contraction = True
        me.init_environments(iprint >= 2)
        dx = DMRG(me, bond_dims, noises)
        dx.noise_type = NoiseTypes.ReducedPerturbative
        dx.davidson_conv_thrds = thrds
        dx.davidson_max_iter, dx.davidson_soft_max_iter = dav_max_iter + 100, dav_max_iter
        dx.iprint, dx.cutoff = iprint, cutoff
        dx.trunc_type = dx.trunc_type | TruncationTypes.RealDensityMatrix
        energy = dx.solve(n_sweeps, ket.center == 0, tol)
        ket.info.bond_dim = max(ket.info.bond_dim, bond_dims[-1])
        self._dmrg = dx
        return energy


def _export():
    """the names above under the pybind module, where block2 keeps them (block2.su2.MovingEnvironment, block2.sz.DMRG, ...;
    src/pybind/pybind_dmrg.hpp:773-900, 1298-1380, 1679-1775), next to the sum-MPO rule classes of parallel.py"""
    from . import b2x_host, parallel

    for sub in (b2x_host.su2, b2x_host.sz):
        for cls in (MovingEnvironment, DMRG, Iteration, MPO, MPS, MPSInfo, ParallelMPO):
            setattr(sub, cls.__name__, cls)
        sub.DMRGEffectiveHamiltonian = EffectiveHamiltonian
        for name in ("ParallelRuleSimple", "ParallelFCIDUMP"):
            setattr(sub, name, getattr(parallel, name))
    for cls in (FuseTypes, NoiseTypes, TruncationTypes, DecompositionTypes, DMRGDriver):
        setattr(b2x_host, cls.__name__, cls)
    return b2x_host


b2x_host = _export()
