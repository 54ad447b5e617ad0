"""Site-to-site DMRG chain on the device: the repo's own two-site sweep loop over block2's symbolic fixtures.

What block2 does per site of a two-site sweep (DMRG::update_two_dot, src/dmrg/sweep_algorithm.hpp:811-1261, around
MovingEnvironment::move_to / eff_ham, src/dmrg/moving_environment.hpp:1534-1570, 2062-2201, and
left/right_contract_rotate, :226-442) is done here with every operator block resident in HBM from the first
environment to the last site:

    blocking   (TensorFunctions::left_contract / right_contract)      -> b2x_outer_build        (element-wise kernel)
    rotation   (TensorFunctions::left_rotate / right_rotate)          -> b2x_plan_create/execute (grouped-GEMM kernels)
    H_eff      (EffectiveHamiltonian ctor: plan + diagonal)           -> b2x_plan_create, b2x_diag_build
    eigs       (IterativeMatrixFunctions::davidson)                   -> device-resident Davidson over b2x_plan_execute
    split      (density matrix of psi, eigenvectors = new MPS tensor) -> host (numpy; a few small symmetric eigenproblems)
    next guess (MovingEnvironment::propagate_wfn, contract_two_dot)   -> host: the wavefunction half of the split regrouped
                                                                         to the next site's fused index (Racah recoupling
                                                                         for SU2) x the neighbouring MPS tensor (_guess)

While the device iterates Davidson on one site, helper threads prepare what the NEXT site needs and what depends on its
structure only: the rotation's GEMM pairs and plan, the blocking terms and their compiled work lists, the walk and the plan of
its effective Hamiltonian, its noise list (_prefetch_next).  Sum-MPO runs either with all ranks in one process (SumMPODMRG) or
with one process per rank and a communicator (DMRG.comm).

The SYMBOLIC side of every step — operator infos, quantum-number bookkeeping, the expressions of the enlarged operators
and of H_eff — belongs to block2's MPO / Partition layers, which are out of scope: it is taken as data from
a chain fixture, i.e. the numbered sequence of blocking / rotation / effective-Hamiltonian events one reference run
recorded (oracle/ref_dump.cpp chain=...; no operator or wavefunction data inside, only the site operators and the MPS
tensors of the starting state).  The NUMERIC side — every operator block of every environment, the wavefunctions, the
MPS tensors after the first decomposition, the energies — is this code's own: nothing numeric is read back from the
reference after the initial environments.  The loop checks that the step it is about to do is the step the reference
did (kind, sweep, site) and otherwise follows its own control flow.
"""
import glob
import os
import re
import time

import numpy as np

from . import capi
from .planfile import OUTER_TERM_DTYPE, PAIR_DTYPE, read_arrays


class ChainFixture:
    """<prefix>.ev<NNN>.<kind>.<ext> files + <prefix>.log of one reference run (ref_dump chain=<sweep> nodelay=1 nocache=1)"""

    def __init__(self, prefix):
        if not os.path.exists(prefix + ".log") and os.path.exists(prefix + ".zip"):
            # a large chain kept as one compressed archive (<name>.zip next to where <name>.log would be): unpacked once
            import tempfile
            import zipfile

            self._tmp = tempfile.TemporaryDirectory(prefix="b2x_chain_")
            with zipfile.ZipFile(prefix + ".zip") as z:
                z.extractall(self._tmp.name)
            prefix = os.path.join(self._tmp.name, os.path.basename(prefix))
        self.events = []
        for fn in sorted(glob.glob(prefix + ".ev*")):
            m = re.search(r"\.ev(\d+)\.([a-z]+)\.e[a-z]+$", fn)
            if m:
                self.events.append((int(m.group(1)), m.group(2), fn))
        self.events.sort()
        self.ref_energy, self.meta = {}, {}
        self.ref_spectra, self.ref_sweep_time, self.ref_total_time = {}, {}, None
        for line in open(prefix + ".log"):
            t = line.split()
            if t and t[0] == "SITE_ENERGY":
                self.ref_energy[(int(t[1]), int(t[2]))] = float(t[3])
            elif t and t[0] == "EVENT":
                self.meta[int(t[1])] = (t[2], int(t[3]), int(t[4]))
            elif t and t[0] == "FINAL_ENERGY":
                self.final_energy = float(t[1])
            elif t and t[0] == "SPECTRA":  # sqrt of every density-matrix eigenvalue the reference truncated at this site
                self.ref_spectra[(int(t[1]), int(t[2]))] = (float(t[3]), np.array([float(x) for x in t[5:]]))
            elif t and t[0] == "SWEEP_TIME":  # the reference's own clock: wall, teff, teig, tprt, tblk, tmve, tdm, tsplt, ...
                self.ref_sweep_time[int(t[1])] = [float(x) for x in t[2:]]
            elif t and t[0] == "TOTAL_TIME":
                self.ref_total_time = float(t[1])
        self.pos = 0

    def preload(self):
        """parse every event file now (block2 holds its MPO and infos in memory; a timed replay should not read them from
        disk inside the clock)"""
        self._mem = {n: read_arrays(fn) for n, _, fn in self.events}
        return self

    def _arrays(self, n, fn):
        mem = getattr(self, "_mem", None)
        d = dict(mem[n]) if mem is not None and n in mem else read_arrays(fn)
        d["_num"] = n  # (the event's number: what the loop's helper thread files its preparations under)
        return d

    def next(self, *kinds):
        n, kind, fn = self.events[self.pos]
        if kind not in kinds:
            raise RuntimeError("chain out of step: the reference did '%s' (event %d), this loop wants %s" % (kind, n, kinds))
        self.pos += 1
        return kind, self._arrays(n, fn)

    def peek(self):
        return self.events[self.pos][1] if self.pos < len(self.events) else None

    def find_next(self, kind):
        """(event number, arrays) of the next event of this kind at or after the cursor, without moving the cursor"""
        for n, k, fn in self.events[self.pos:]:
            if k == kind:
                return n, self._arrays(n, fn)
        return None, None


def host_cores():
    """cores this process may actually use: the affinity mask, cut by the cgroup CPU quota (a container on a 256-thread host
    with a 16-core quota sees 256 CPUs)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


_blas_limited = False


def limit_host_blas():
    """The host steps of the loop (the density-matrix eigenproblems of a split, the small products of the carried
    wavefunction) run on numpy's BLAS, whose pool is sized from the CPUs it SEES: on the GPU box that is 64 threads inside a
    16-core quota, and the sector-sized eigh calls of a split ran 6-8x slower than on one thread, with 10x jitter
    (tools/blas_threads_probe.py; profiles/r03_host_blas_threads.txt).  Once per process the pool is cut to
    min(cores available, 8); B2X_HOST_BLAS_THREADS overrides (0 = leave the pool alone)."""
    global _blas_limited
    if _blas_limited:
        return
    _blas_limited = True
    want = os.environ.get("B2X_HOST_BLAS_THREADS")
    n = int(want) if want is not None else min(host_cores(), 8)
    if n <= 0:
        return
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        return
    threadpool_limits(limits=n, user_api="blas")


_pool_shared = None


def _helper_pool():
    """the helper threads of the sweep loops of this process (DMRG._prefetch_next).  Two: the walks and the compilers run
    without the GIL, so the rotation + blockings of the next site and its effective Hamiltonian proceed side by side (each
    result is waited for by its own event number).  B2X_SWEEP_HELPERS sets the number."""
    global _pool_shared
    if _pool_shared is None:
        from concurrent.futures import ThreadPoolExecutor

        _pool_shared = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("B2X_SWEEP_HELPERS", "2"))),
                                          thread_name_prefix="b2x-prefetch")
    return _pool_shared


_tp_controller = None


class _one_blas_thread:
    """context: numpy's BLAS / LAPACK on one thread (threadpoolctl; a no-op without it or with B2X_HOST_BLAS_THREADS=0)"""

    def __enter__(self):
        self._ctx = None
        if os.environ.get("B2X_HOST_BLAS_THREADS") == "0":
            return self
        global _tp_controller
        try:
            if _tp_controller is None:  # (made once: a controller scans the loaded libraries, 0.6 ms)
                from threadpoolctl import ThreadpoolController

                _tp_controller = ThreadpoolController()
            self._ctx = _tp_controller.limit(limits=1, user_api="blas")
        except Exception:
            return self
        self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)
        return False


_heap_retained = False


def retain_host_heap():
    """The host walks of a site (blocking, rotation, H_eff) build lists of 1e5..1e6 records, tens of MB each, in memory that
    glibc maps and unmaps per allocation above its mmap threshold — and a page fault costs microseconds in a virtualised
    container: filling the 26 MB record array of a Cr2 M=250 blocking took 100 ms of page faults against 6 ms of copying
    (B2X_PLAN_DEBUG=1 clocks, tools/outer_build_clock.py).  Once per process the thresholds are raised so that freed blocks stay
    in the heap and are reused (mallopt: M_MMAP_THRESHOLD 1 GiB, M_TRIM_THRESHOLD 2 GiB, M_TOP_PAD 64 MiB); B2X_HOST_HEAP=0
    leaves the allocator alone."""
    global _heap_retained
    if _heap_retained or os.environ.get("B2X_HOST_HEAP", "1") == "0":
        return
    _heap_retained = True
    try:
        import ctypes

        libc = ctypes.CDLL(None)
        libc.mallopt(-3, 1 << 30)        # M_MMAP_THRESHOLD
        libc.mallopt(-1, (2 << 30) - 1)  # M_TRIM_THRESHOLD
        libc.mallopt(-2, 64 << 20)       # M_TOP_PAD
    except (OSError, AttributeError):
        pass


class OpTensor:
    """operator blocks of one (enlarged or rotated) block in HBM: one device vector + {operator key: (offset, length)}"""

    def __init__(self, n, layout):
        self.buf = capi.DeviceBuffer(max(int(n), 1))
        self.n, self.layout = int(n), layout

    def close(self):
        self.buf.close()


def _info(d, i):
    memo = d.setdefault("_info_memo", {})  # (an event names the same info many times: once per operator)
    v = memo.get(int(i))
    if v is None:
        pre = "info.%d." % int(i)
        q = d[pre + "quanta"].astype(np.uint64)
        v = memo[int(i)] = {"q": q, "nbra": d[pre + "nbra"].astype(np.int64), "nket": d[pre + "nket"].astype(np.int64),
                            "ntot": d[pre + "ntot"].astype(np.int64), "dq": int(d[pre + "meta"][0]),
                            "len": int(d[pre + "meta"][3])}
    return v


def _address_space(n):
    """a host array a recording walk takes ADDRESSES from (operator blocks, psi): never read, never written, so not
    initialised either (np.zeros would have to clear it once the heap is retained, retain_host_heap)"""
    return np.empty(max(int(n), 1))[:int(n)]


def _records(raw, dtype):
    """the byte array a symbolic walk returns, seen as its records (no copy: a blocking list is tens of MB)"""
    a = np.asarray(raw)
    if a.dtype == np.uint8 and a.flags.c_contiguous and a.ctypes.data % 8 == 0:
        return a.view(dtype)
    return np.frombuffer(a.tobytes(), dtype)


def _fields(q):
    """packed SU2 / SZ label -> (n, twos_low, twos, pg); n is a signed 16-bit field (src/core/symmetry.hpp:1183-1306)"""
    q = np.asarray(q, np.uint64)
    n = ((q >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64)
    n = np.where(n >= 32768, n - 65536, n)
    return (n, ((q >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64),
            ((q >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64), (q & np.uint64(0xFFFF)).astype(np.int64))


class Timers(dict):
    def add(self, k, t0):
        self[k] = self.get(k, 0.0) + (time.perf_counter() - t0)


class DMRG:
    """two-site sweeps over a chain fixture; `sym` = "su2" (labels carry twos_low) or "sz".  Usage:
        dm = DMRG(ChainFixture(prefix), "su2"); dm.init_environments(); e0 = dm.sweep(0, True); e1 = dm.sweep(1, False)"""

    def __init__(self, fixture, sym, conv_thrd=1e-13, seed=1234):
        from . import b2x_host

        limit_host_blas()
        retain_host_heap()
        self.fx, self.sym, self.host = fixture, sym, b2x_host
        self.conv_thrd, self.rng = conv_thrd, np.random.default_rng(seed)
        self.L, self.R = {}, {}          # rotated blocks by the site their enlarged successor starts from
        self.EL = self.ER = None         # current enlarged blocks
        self.psi = None                  # (host wavefunction, its info, forward?) of the last solved site
        self.energies, self.tm, self.n_sites = {}, Timers(), None
        self.ndav = {}
        self.n_zero_ops, self.zero_log = 0, []
        self.pket = None                 # perturbed wavefunctions of the last solved site (noisy sweeps)
        self.check_truncation = False    # True: every split also makes its OWN choice of kept states and logs where it
        self.trunc_log = {}              # differs from the fixture's bond dimensions (see _split)
        self.site_key = None
        self.ahead = None                # (event number, new MPS tensor): a split made before the move (split_site)
        self.last_split = None           # {"error": discarded weight, "mmps": kept states} of the last split
        # the state carried from site to site (block2: MPS::tensors / MPSInfo): the MPS tensors on both sides of the centre,
        # the site bases, and the wavefunction half of the last split — what the next site's Davidson starts from (_guess)
        self.use_previous = True         # False: every site starts from the low end of the diagonal (the round-2 behaviour)
        # sum-MPO with one rank per PROCESS: a parallel.ParallelCommunicator (RCCL, or gloo for ranks that share a card).  This
        # engine then carries ITS rank's event chain: sigma, the diagonal and the perturbed wavefunctions are summed over the
        # ranks (ParallelTensorFunctions::operator(), parallel_tensor_functions.hpp:51-55; effective_hamiltonian.hpp:399-400),
        # every rank runs the same Davidson and the same split on the same numbers
        self.comm = None
        self.basis, self.mpsL, self.mpsR = {}, {}, {}
        self.carry, self._at = None, None
        # the next site's effective Hamiltonian prepared on a helper thread while the device solves this site (_prefetch_next)
        self.prefetch = os.environ.get("B2X_SWEEP_PREFETCH", "1") != "0"
        self._pool, self._ahead, self.n_prefetched, self.prefetch_errors = None, {}, 0, []
        self.guess_log = {}              # (sweep, site) -> (how the starting vector was made: "previous" / "same" / "diagonal",
                                         #                   its overlap with the solution)

    # ---- steps ------------------------------------------------------------------------------------------------
    def _assign(self, d):
        """first block of a chain: the enlarged block IS the site operator tensor (left_assign / right_assign)"""
        t0 = time.perf_counter()
        site = np.ascontiguousarray(d["site"], np.float64)
        bkey = {int(k): (int(o), int(l)) for k, o, l in zip(d["b.key"], d["b.off"], d["b.len"]) if o >= 0}
        layout, host = {}, np.zeros(int(d["meta"][4]))
        for k, i, o in zip(d["c.key"], d["c.info"], d["c.off"]):
            if o < 0:
                continue
            n = _info(d, i)["len"]
            so, sl = bkey[int(k)]
            assert sl == n
            host[o:o + n] = site[so:so + n]
            layout[int(k)] = (int(o), n)
        t = OpTensor(len(host), layout)
        t.buf.upload(host)
        self.tm.add("assign", t0)
        return t

    def _repack(self, src, keys, offs, lens, total):
        """device vector of `total` elements holding operator `key` at `off` (the layout a fixture's step expects); the
        source tensor is returned as it is when it already has that layout (the usual case: producer and consumer saw
        the same memory in the reference)"""
        want = {int(k): (int(o), int(l)) for k, o, l in zip(keys, offs, lens) if o >= 0 and l > 0}
        if total == src.n and all(src.layout.get(k) == v for k, v in want.items()):
            return src, False
        out = OpTensor(total, want)
        do, so_, ln = [], [], []
        for k, (o, l) in want.items():
            if k not in src.layout:  # allocated by the reference but never written (a symbol that is zero at this site)
                self.n_zero_ops += 1
                self.zero_log.append((self.fx.pos - 1, k, l))
                continue
            so, sl = src.layout[k]
            assert sl == l, "operator %x: length %d here, %d expected" % (k, sl, l)
            do.append(o), so_.append(so), ln.append(l)
        capi.gather_d2d(out.buf.ptr, src.buf.ptr, do, so_, ln)
        return out, True

    def _block(self, d, blk):
        """enlarged block = block (x) site: symbolic_blocking records the element-wise terms, the device executes them"""
        t0 = time.perf_counter()
        ready = self._take_ready()
        right = bool(d["meta"][2])
        pre = "rop" if right else "lop"  # the block-operator side of tensor_product's (lop, rop)
        self._last_basis = self._site_basis(d, "lop" if right else "rop")
        xl = int(d["x.len"][0])
        lens = [(_info(d, i)["len"] if o >= 0 else 0) for i, o in zip(d[pre + ".info"], d[pre + ".off"])]
        x, tmp = self._repack(blk, d[pre + ".key"], d[pre + ".off"], lens, xl)
        terms, vlen, sum_terms, tmp_len = ready[:4] if ready is not None else self._record_blocking(d)
        op = ready[4] if ready is not None else None
        layout = {int(k): (int(o), _info(d, i)["len"]) for k, i, o in zip(d["c.key"], d["c.info"], d["c.off"])}
        out = OpTensor(vlen, layout)
        site = capi.Arena.from_host([np.ascontiguousarray(d["site"], np.float64)])
        x_len = xl
        if sum_terms is not None:
            # operator sums with transposed members (sum-MPO MPOs): the temporaries live behind the block operators in the
            # input vector and are formed first, by a pass that reads and writes that extended vector
            x_len = xl + tmp_len
            xe = OpTensor(x_len, {})
            if xl:
                capi.memcpy_d2d(xe.buf.ptr, x.buf.ptr, xl)
            if tmp:
                x.close()
            x, tmp = xe, True
            capi.outer_build(site, sum_terms, x.buf.ptr, x.buf.ptr, True, x_len, x_len)
        if op is not None:
            op.execute(site, x.buf.ptr, out.buf.ptr)
        else:
            capi.outer_build(site, terms, x.buf.ptr, out.buf.ptr, True, x_len, vlen)
        capi.device_sync()
        if op is not None:
            op.close()
        site.close()
        if tmp:
            x.close()
        self.tm.add("block", t0)
        return out

    def _site_basis(self, d, pre):
        """StateInfo of the site of a blocking event (block2: MPSInfo::basis[i]): the labels of a site operator without a
        quantum-number change list every site state once, in block2's order -> ([(n, 2S or 2Sz, pg)], [n_states])"""
        best = None
        for i in sorted(set(int(x) for x in d[pre + ".info"])):
            inf = _info(d, i)
            if inf["dq"] == 0 and (best is None or len(inf["q"]) > len(best["q"])):
                best = inf
        if best is None:
            return None
        n, _, tw, pg = _fields(best["q"])
        if self.sym == "sz":
            tw = np.where(tw >= 32768, tw - 65536, tw)
        return [(int(a), int(b), int(c)) for a, b, c in zip(n, tw, pg)], [int(x) for x in best["nket"]]

    def _rotate(self, d, enl, mps):
        """rotated block = A^T . enlarged . A per operator sector: symbolic_rotate records the GEMM pairs"""
        t0 = time.perf_counter()
        pairs = self._take_ready()
        xl, vl, al = int(d["meta"][5]), int(d["meta"][6]), int(d["meta"][7])
        lens = [_info(d, i)["len"] for i in d["a.info"]]
        x, tmp = self._repack(enl, d["a.key"], d["a.off"], lens, xl)
        if pairs is None:
            pairs = self._record_rotation(d)
        assert len(mps) == al
        arena = capi.Arena.from_host([np.ascontiguousarray(mps, np.float64)])
        plan = capi.Plan(arena, pairs, xl, vl)
        assert plan.stats["fallback"] == 0
        layout = {int(k): (int(o), _info(d, i)["len"]) for k, i, o in zip(d["c.key"], d["c.info"], d["c.off"])}
        out = OpTensor(vl, layout)
        plan.execute_device(x.buf.ptr, out.buf.ptr, 1.0)
        capi.device_sync()
        plan.close(), arena.close()
        if tmp:
            x.close()
        self.tm.add("rotate", t0)
        return out

    def _transform(self, d, rot):
        """the NC -> CN switch of the conventional MPO at the middle of the chain: the rotated block gains new
        (complementary) operators that are sums of its own operators or their transposes
        (TensorFunctions::numerical_transform); the sums run on the device, in place in the block's vector"""
        t0 = time.perf_counter()
        ready = self._take_ready(d.get("_num", -1))
        total = int(d["meta"][3])
        lens = [(_info(d, i)["len"] if o >= 0 else 0) for i, o in zip(d["t.info"], d["t.off"])]
        have = {int(k): (int(o), int(l)) for k, o, l in zip(d["t.key"], d["t.off"], lens) if o >= 0 and l > 0}
        out = OpTensor(total, have)  # zero-filled: the new operators start from zero
        do, so_, ln = [], [], []
        for k, (o, l) in have.items():
            if k in rot.layout:
                so, sl = rot.layout[k]
                assert sl == l
                do.append(o), so_.append(so), ln.append(l)
        capi.gather_d2d(out.buf.ptr, rot.buf.ptr, do, so_, ln)
        dummy = capi.Arena.from_host([np.zeros(1)])
        if ready is not None:
            ready[1].execute(dummy, out.buf.ptr, out.buf.ptr)
        else:
            terms = _records(self.host.symbolic_transform(self.sym, d), OUTER_TERM_DTYPE)
            capi.outer_build(dummy, terms, out.buf.ptr, out.buf.ptr, True, total, total)
        capi.device_sync()
        if ready is not None:
            ready[1].close()
        dummy.close(), rot.close()
        self.tm.add("transform", t0)
        return out

    def _rotate_and_transform(self, d, enl, mps):
        t = self._rotate(d, enl, mps)
        while self.fx.peek() in ("lntr", "rntr", "lint", "rint"):  # operator sums formed inside the rotated block
            t = self._transform(self.fx.next("lntr", "rntr", "lint", "rint")[1], t)
        return t

    def _eff_ham(self, d):
        """H_eff of the site from the two enlarged blocks (all in HBM): the plan and THIS Hamiltonian's diagonal on the
        device.  (sum-MPO: one such part per rank; the parts are solved together, _solve)"""
        t0 = time.perf_counter()
        al = int(d["arena.len"][0])
        ready = self._take_prefetched()
        if ready is not None:
            pairs, dterms = ready
        else:
            pairs, dterms = self._record_eff_ham(d)
        kinfo = _info(d, d["ket.info"][0])
        n = kinfo["len"]
        self.tm.add("eff_ham.record", t0)
        t0 = time.perf_counter()
        arena_t = OpTensor(al, {})
        for pre, blk in (("lopt", self.EL), ("ropt", self.ER)):
            do, so_, ln = [], [], []  # the operators of one enlarged block gathered into the arena in one launch
            for k, o, l in zip(d[pre + ".key"].tolist(), d[pre + ".off"].tolist(), d[pre + ".len"].tolist()):
                if o >= 0 and l > 0:
                    if k not in blk.layout:
                        self.n_zero_ops += 1
                        self.zero_log.append((self.fx.pos - 1, k, l))
                        continue
                    so, sl = blk.layout[k]
                    assert sl == l
                    do.append(o), so_.append(so), ln.append(l)
            capi.gather_d2d(arena_t.buf.ptr, blk.buf.ptr, do, so_, ln)
        arena = capi.Arena.adopt_device(arena_t.buf.ptr, al, keep=arena_t)
        plan = capi.Plan(arena, pairs, n, n)
        assert plan.stats["fallback"] == 0
        diag = capi.DeviceBuffer(n)
        capi.diag_build(arena, dterms, diag.ptr, n, True)
        self.tm.add("eff_ham.device", t0)
        return {"plan": plan, "arena": arena, "arena_t": arena_t, "diag": diag, "kinfo": kinfo, "n": n,
                "const_e": float(d["const_e"][0]), "n_pairs": len(pairs)}

    def _record_eff_ham(self, d):
        """the symbolic walk of one effective Hamiltonian: (GEMM pairs, diagonal terms); needs no device and no operator data"""
        dd = dict(d)
        dd["arena"] = _address_space(int(d["arena.len"][0]))
        h = self.host.SymbolicEffectiveHamiltonian(self.sym, dd)
        h.record()
        return _records(h.pairs(), PAIR_DTYPE), np.asarray(h.diag_terms())

    # The host work of a site that depends on the STRUCTURE of the next site only — the walk of its effective Hamiltonian and
    # the compilation of its H.psi plan, the largest single host cost of a site below M ~ 1000 — runs on a helper thread
    # while the device iterates Davidson on the current site (davidson_device releases the GIL; ctypes calls do anyway).  The
    # helper creates the plan over a scratch arena of the right extent and destroys it: the library parks a destroyed plan in
    # its compiled-plan cache (b2x_plan_destroy), and the plan creation of the main thread takes it from there and only binds
    # it to the real arena.  Nothing changes when the helper is late, fails or is switched off (B2X_SWEEP_PREFETCH=0): the
    # main thread then walks and compiles itself, as before.
    _PREFETCH_MAX_ARENA = 1 << 30  # elements (8 GB): above that the device time of a site dwarfs the compilation anyway

    def _prefetch_next(self):
        """queue the structure-only work of every step between here and the next effective Hamiltonian (inclusive): the
        rotation's GEMM pairs + its plan (primed into the plan cache), the element-wise terms of the blockings, the walk +
        plan of the effective Hamiltonian"""
        if not self.prefetch:
            return
        fx = self.fx
        todo = []
        for k in range(fx.pos, len(fx.events)):
            num, kind, fn = fx.events[k]
            if kind in ("lblk", "rblk") and k + 1 < len(fx.events) and fx.events[k + 1][1] in ("lrot", "rrot"):
                continue  # (the re-contraction block2 does before a rotation: the enlarged block is still in HBM here)
            if kind in ("lrot", "rrot", "lblk", "rblk", "eham", "enoise", "lntr", "rntr", "lint", "rint") and num not in self._ahead:
                todo.append((num, kind, fn))
            if kind == "eham":  # (+ the perturbative-noise step of the same site, on a noisy sweep)
                if k + 1 < len(fx.events) and fx.events[k + 1][1] == "enoise" and fx.events[k + 1][0] not in self._ahead:
                    todo.append(fx.events[k + 1])
                break
        if not todo:
            return
        if self._pool is None:
            self._pool = _helper_pool()
        self._ahead = {n: f for n, f in self._ahead.items() if not f.done() or n >= todo[0][0]}  # (drop what was never taken)
        for num, kind, fn in todo:
            self._ahead[num] = self._pool.submit(self._prepare, num, kind, fn, capi.current_device())

    def _prepare(self, num, kind, fn, ordinal):
        try:
            capi.device_init(ordinal)  # (the device is selected per thread)
            d = self.fx._arrays(num, fn)
            if kind == "eham":
                al = int(d["arena.len"][0])
                if al > self._PREFETCH_MAX_ARENA:
                    return None
                pairs, dterms = self._record_eff_ham(d)
                n = _info(d, d["ket.info"][0])["len"]
                tmp = capi.DeviceBuffer(al)
                arena = capi.Arena.adopt_device(tmp.ptr, al, keep=tmp)
                plan = capi.Plan(arena, pairs, n, n)
                plan.close(), arena.close(), tmp.close()
                return pairs, dterms
            if kind == "enoise":
                al = int(d["arena.len"][0])
                if al > self._PREFETCH_MAX_ARENA:
                    return None
                gemms, n, out_len = self._record_noise(d)
                tmp = capi.DeviceBuffer(al)
                arena = capi.Arena.adopt_device(tmp.ptr, al, keep=tmp)
                gp = capi.GemmPlan(arena, gemms, n, out_len)
                gp.close(), arena.close(), tmp.close()
                return gemms
            if kind in ("lrot", "rrot"):
                xl, vl, al = int(d["meta"][5]), int(d["meta"][6]), int(d["meta"][7])
                pairs = self._record_rotation(d)
                arena = capi.Arena.from_host([np.zeros(al)])
                plan = capi.Plan(arena, pairs, xl, vl)
                plan.close(), arena.close()
                return pairs
            if kind in ("lntr", "rntr", "lint", "rint"):  # operator sums inside a rotated block: terms + compiled work list
                total = int(d["meta"][3])
                terms = _records(self.host.symbolic_transform(self.sym, d), OUTER_TERM_DTYPE)
                return terms, capi.OuterPlan(terms, 1, total, total)
            terms, vlen, sum_terms, tmp_len = self._record_blocking(d)
            # the work list of the block products compiled and uploaded now (the site operators are the arena of this step)
            op = capi.OuterPlan(terms, len(d["site"]), int(d["x.len"][0]) + tmp_len, vlen)
            return terms, vlen, sum_terms, tmp_len, op
        except Exception as e:  # the main thread does the work itself (and reports its own error, if it has one too)
            self.prefetch_errors.append((num, kind, repr(e)))
            return None

    def _take_ready(self, num=None):
        """what the helper prepared for the event the cursor just passed (or event `num`), or None"""
        if num is None:
            num = self.fx.events[self.fx.pos - 1][0]
        fut = self._ahead.pop(num, None)
        if fut is None:
            return None
        res = fut.result()
        if res is not None:
            self.n_prefetched += 1
        return res

    def _take_prefetched(self):
        return self._take_ready(getattr(self, "_eham_num", None))

    def _record_noise(self, d):
        """the single-GEMM list of the perturbative-noise step -> (records, psi length, length of the perturbed wavefunctions)"""
        from .planfile import GEMM_DTYPE

        al, n = int(d["arena.len"][0]), _info(d, d["ket.info"][0])["len"]
        dd = dict(d)
        dd["arena"], dd["psi"] = _address_space(al), _address_space(n)
        h = self.host.SymbolicEffectiveHamiltonian(self.sym, dd)
        gb, _ = h.perturbative_noise(dd, False)
        return np.frombuffer(bytes(gb), GEMM_DTYPE), n, int(d["noise.args"][5])

    def _record_rotation(self, d):
        dd = dict(d)
        dd["x"], dd["arena"] = _address_space(int(d["meta"][5])), _address_space(int(d["meta"][7]))
        pairs, _ = self.host.symbolic_rotate(self.sym, dd, False)
        return _records(pairs, PAIR_DTYPE)

    def _record_blocking(self, d):
        """-> (terms, length of the enlarged block's vector, terms of the temporaries or None, length of their area)"""
        dd = dict(d)
        dd["x"] = _address_space(int(d["x.len"][0]))
        res = self.host.symbolic_blocking(self.sym, dd, False)
        terms, vlen = _records(res[0], OUTER_TERM_DTYPE), len(res[1])
        if len(res) == 4:
            return terms, vlen, _records(res[2], OUTER_TERM_DTYPE), int(res[3])
        return terms, vlen, None, 0

    def _perturb(self, d, part, ket):
        """EffectiveHamiltonian::perturbative_noise (src/dmrg/effective_hamiltonian.hpp:252-423) on the device: the symbolic
        walk records the single-GEMM list (host mirror, from the fixture's sub-labels and perturbed-ket infos), the list runs
        on the grouped-GEMM kernel over the SAME operator arena as H.psi (b2x_gemm_plan_create), the perturbed
        wavefunctions come back to the host for the density matrix.  d = the site's `enoise` event."""
        t0 = time.perf_counter()
        al, n = int(d["arena.len"][0]), part["n"]
        assert al == part["arena_t"].n, "the noise step of a site shares the operator arena of its effective Hamiltonian"
        out_len = int(d["noise.args"][5])
        gemms = self._take_ready(d.get("_num", -1))
        if gemms is None:
            gemms, n_, out_len = self._record_noise(d)
            assert n_ == n
        self.tm.add("noise.record", t0)
        t0 = time.perf_counter()
        gp = capi.GemmPlan(part["arena"], gemms, n, out_len)
        out = capi.DeviceBuffer(out_len)
        gp.execute_device(ket.ptr, out.ptr, 1.0)
        capi.device_sync()
        if self.comm is not None and self.comm.size > 1:  # comm->reduce_sum(perturb_ket, root): here every rank keeps the sum
            self.comm.allreduce_device(out.ptr, out_len)
        pk = out.download()
        gp.close(), out.close()
        infos = [_info(d, i) for i in d["noise.vinfo"]]
        self.tm.add("noise.device", t0)
        return {"data": pk, "infos": infos, "offs": [int(o) for o in d["noise.voff"]], "noise": float(d["noise.value"][0]),
                "n_gemms": len(gemms)}

    def _solve(self, parts, noise_event=None):
        """Davidson on the device over H = sum of the parts' plans (one part: the serial case; several: the sum-MPO
        Hamiltonian H = sum_r H_r with every H_r psi accumulated into the same sigma — on separate GPUs that sum is the
        all-reduce of ParallelTensorFunctions::operator()).  The diagonals of the parts are summed likewise."""
        import ctypes as C

        t0 = time.perf_counter()
        p0 = parts[0]
        n, diag, plan = p0["n"], p0["diag"], p0["plan"]
        for q in parts[1:]:
            assert q["n"] == n
            capi.check(capi.lib().b2x_vec_axpy(C.c_double(1.0), C.c_void_p(q["diag"].ptr), C.c_void_p(diag.ptr), C.c_size_t(n), None))
        more = [q["plan"]._h.value for q in parts[1:]]
        comm = self.comm if self.comm is not None and self.comm.size > 1 else None
        if comm is not None:
            comm.allreduce_device(diag.ptr, n)  # (the diagonal of H = sum_r H_r)
        # Initial guess.  block2 starts Davidson from the wavefunction of the previous site moved to this one
        # (MovingEnvironment::propagate_wfn + contract_two_dot); so does this loop (_guess).  Where there is no previous
        # wavefunction (the first site of a calculation) it starts from the low end of the diagonal, the usual Davidson
        # guess.  Either way the answer is checked against the variational bound E0 <= min(diag): a Ritz pair with a tiny
        # residual above that bound is an excited state the iteration fell into, and the site is solved again from the
        # lowest diagonal entry.
        dg = diag.download()
        guess, how = self._guess(p0["kinfo"]) if self.use_previous else (None, None)
        if guess is None:
            guess, how = 1.0 / (dg - dg.min() + 0.1) ** 2 + 1e-3 * self.rng.standard_normal(n), "diagonal"
        ket = capi.DeviceBuffer(n, guess)
        ndav = 0
        self._prefetch_next()
        for attempt in range(4):
            e, nd = self.host.davidson_device(plan._h.value, diag.ptr, ket.ptr, n, self.conv_thrd, 5000, more_plans=more,
                                              comm=self._davidson_comm(comm))
            ndav += nd
            if e <= dg.min() + 1e-9:
                break
            guess = 1e-2 * self.rng.standard_normal(n)
            guess[np.argsort(dg)[:attempt + 1]] += 1.0
            ket.upload(guess)
        psi = ket.download()
        self._guess_how = (how, float(abs(guess @ psi) / max(np.linalg.norm(guess) * np.linalg.norm(psi), 1e-300)))
        self.tm.add("eigs", t0)
        if os.environ.get("B2X_SWEEP_DEBUG"):  # residual of the returned pair, and the diagonal as the kernels built it
            sig = capi.DeviceBuffer(n)
            for q in parts:
                q["plan"].execute_device(ket.ptr, sig.ptr, 1.0)
            capi.device_sync()
            hs = sig.download()
            rq = float(psi @ hs) / float(psi @ psi)
            print("   [debug] n=%d parts=%d ndav=%d e=%.10f rayleigh=%.10f |r|=%.2e |psi|=%.6f diag[min,max]=%.3f,%.3f" % (
                n, len(parts), ndav, e, rq, np.linalg.norm(hs - rq * psi), np.linalg.norm(psi), dg.min(), dg.max()), flush=True)
            sig.close()
        const_e = parts[0]["const_e"]  # (every rank's fixture carries the SAME constant: it is added once, by the root)
        self.const_e = const_e
        self.pket = None
        if noise_event is not None:
            # sum-MPO: every rank perturbs psi with ITS operators into the same layout (the perturbed labels are all-reduced
            # before the infos are made, effective_hamiltonian.hpp:303-309) and the results are summed on the root
            # (comm->reduce_sum, :399-400, 416-417); here the ranks' lists run one after another and are added
            events = noise_event if isinstance(noise_event, (list, tuple)) else [noise_event]
            if len(events) != len(parts):
                raise RuntimeError("%d noise events for %d parts of the Hamiltonian" % (len(events), len(parts)))
            for ev, q in zip(events, parts):
                pk = self._perturb(ev, q, ket)
                if self.pket is None:
                    self.pket = pk
                else:
                    if pk["offs"] != self.pket["offs"] or len(pk["data"]) != len(self.pket["data"]):
                        raise RuntimeError("the ranks' perturbed wavefunctions differ in layout")
                    self.pket["data"] += pk["data"]
                    self.pket["n_gemms"] += pk["n_gemms"]
        for q in parts:
            q["plan"].close(), q["arena"].close(), q["arena_t"].close(), q["diag"].close()
        ket.close()
        return e + const_e, ndav, psi, p0["kinfo"], sum(q["n_pairs"] for q in parts)

    @staticmethod
    def _davidson_comm(comm):
        """what davidson_device takes: the C ABI's RCCL communicator as (handle, rank, size, root), any other transport as the
        communicator object itself (its allreduce_device / broadcast_device are called back)"""
        if comm is None:
            return None
        if getattr(comm, "_rccl", None) is not None:
            return (comm._rccl._h.value, comm.rank, comm.size, comm.root)
        return comm

    def _eigs(self, d, noise_event=None):
        return self._solve([self._eff_ham(d)], noise_event)

    # ---- the wavefunction carried to the next site ----------------------------------------------------------------
    def _fuse(self, a, b):
        """labels of the product of two states (S::operator+): SU2 couples |Sa - Sb| .. Sa + Sb, SZ adds"""
        if self.sym == "sz":
            return [(a[0] + b[0], a[1] + b[1], a[2] ^ b[2])]
        return [(a[0] + b[0], t, a[2] ^ b[2]) for t in range(abs(a[1] - b[1]), a[1] + b[1] + 1, 2)]

    def _connection(self, ak, ad, bk, bd):
        """StateInfo::get_connection_info (src/core/state_info.hpp:283-311): for every fused label the (i, j) pairs of a (x) b
        that land in it, first index outermost, and where each pair's a_i * b_j states start -> {label: [width, {(i, j): start}]}"""
        out = {}
        for i, qa in enumerate(ak):
            for j, qb in enumerate(bk):
                for q in self._fuse(qa, qb):
                    e = out.setdefault(q, [0, {}])
                    e[1][(i, j)] = e[0]
                    e[0] += ad[i] * bd[j]
        return out

    def _recoupling(self, ta, tb, tc, td, te, tf):
        """the factor of SparseMatrix::swap_to_fused_left / _right (src/core/sparse_matrix.hpp:1838-1843, 1911-1916):
        racah(a, b, c, d, e, f) sqrt((2e + 1)(2f + 1)), racah = (-1)^(a+b+c+d) {a b e; d c f} (clebsch_gordan.hpp:177-180);
        1 for SZ"""
        if self.sym == "sz":
            return 1.0
        memo = self.__dict__.setdefault("_racah_memo", {})
        k = (ta, tb, tc, td, te, tf)
        v = memo.get(k)
        if v is None:
            v = memo[k] = (1 - ((ta + tb + tc + td) & 2)) * self.host.wigner_6j(ta, tb, te, td, tc, tf) * np.sqrt(
                (te + 1.0) * (tf + 1.0))
        return v

    def _guess(self, kinfo):
        """Davidson's starting vector for the site the environments were just moved to (self._at): the wavefunction of the
        previous site carried over as block2 does.  After the split psi = L . (S V^T) of a forward step the wavefunction
        half, a matrix [bond] x [fused (site, right bond)], is regrouped to [fused (bond, site)] x [right bond]
        (MPSInfo::swap_wfn_to_fused_left, src/dmrg/mps.hpp:715-742; SU2: one Racah coefficient per pair of coupling
        paths) and multiplied with the MPS tensor of the site to its right (MovingEnvironment::contract_two_dot,
        src/dmrg/moving_environment.hpp:3319-3362, SparseMatrix::contract, sparse_matrix.hpp:1742-1786); the backward
        step mirrors it.  The bond and site StateInfos are this loop's own (its split, the tensors it rotated with, the
        site operators' labels); a structure it cannot match returns (None, None) and the caller falls back.
        -> (vector in kinfo's layout, "previous" | "same") or (None, None)"""
        t0 = time.perf_counter()
        try:
            return self._guess_impl(kinfo)
        finally:
            self.tm.add("guess", t0)

    def _guess_impl(self, kinfo):
        i, forward = self._at
        c = self.carry
        if c is None:
            # the turn-around of a sweep: the same two sites again, in the same bases
            if (self.psi is not None and self.site_key is not None and self.site_key[1] == i
                    and len(self.psi[0]) == kinfo["len"] and np.array_equal(self.psi[1]["q"], kinfo["q"])
                    and np.array_equal(self.psi[1]["nbra"], kinfo["nbra"]) and np.array_equal(self.psi[1]["nket"], kinfo["nket"])):
                return self.psi[0].copy(), "same"
            return None, None
        if c["forward"] != forward or c["site"] != (i - 1 if forward else i + 1):
            return None, None
        _, _, ttw, _ = _fields(np.array([kinfo["dq"]], np.uint64))
        ttw = int(ttw[0])
        ln, ltw, lpg = self._bond_labels(kinfo, False)
        rn, rtw, rpg = self._bond_labels(kinfo, True)
        out = np.zeros(kinfo["len"])
        blocks = c["blocks"]
        if forward:
            # previous sites (i - 1, i): blocks[(a, mr)] = [a] x [fused (m_i, r_{i+1})]; now [fused (a, m_i)] x [r_{i+1}] . R_{i+1}
            t, m = self.mpsR.get(i + 1), self.basis.get(i)
            la = self.mpsL.get(i)
            if t is None or m is None or la is None:
                return None, None
            ak, ad = la["keys"], [int(x) for x in la["info"]["nket"]]
            mk, md = m
            rk, rd = t["keys"], [int(x) for x in t["info"]["nbra"]]
            lm, mr = self._connection(ak, ad, mk, md), self._connection(mk, md, rk, rd)
            rpos = {q: j for j, q in enumerate(rk)}
            for b in range(len(ln)):
                ql, qr = (int(ln[b]), int(ltw[b]), int(lpg[b])), (int(rn[b]), int(rtw[b]), int(rpg[b]))
                nb, nk = int(kinfo["nbra"][b]), int(kinfo["nket"][b])
                ic = rpos.get(qr)
                if ic is None or ql not in lm:
                    continue
                ti = t["info"]
                if lm[ql][0] != nb or int(ti["nket"][ic]) != nk:
                    return None, None
                w = np.zeros((nb, rd[ic]))
                for (ia, im), off in lm[ql][1].items():
                    rows = ad[ia] * md[im]
                    for qmr in self._fuse(mk[im], rk[ic]):
                        src = blocks.get((ak[ia], qmr))
                        if src is None:
                            continue
                        if src.shape[1] != mr[qmr][0]:
                            return None, None
                        p = mr[qmr][1][(im, ic)]
                        f = self._recoupling(ak[ia][1], mk[im][1], ttw, rk[ic][1], ql[1], qmr[1])
                        w[off:off + rows] += f * src[:, p:p + md[im] * rd[ic]].reshape(rows, rd[ic])
                o = t["base"] + int(ti["ntot"][ic])
                rblk = t["data"][o:o + rd[ic] * nk].reshape(rd[ic], nk)
                o = int(kinfo["ntot"][b])
                out[o:o + nb * nk] = (w @ rblk).reshape(-1)
        else:
            # previous sites (i + 1, i + 2): blocks[(lm, a)] = [fused (l_{i+1}, m_{i+1})] x [a]; now L_i . [l_{i+1}] x [fused (m_{i+1}, a)]
            t, m = self.mpsL.get(i + 1), self.basis.get(i + 1)
            ra = self.mpsR.get(i + 2)
            if t is None or m is None or ra is None:
                return None, None
            ak, ad = ra["keys"], [int(x) for x in ra["info"]["nbra"]]
            mk, md = m
            lk, ld = t["keys"], [int(x) for x in t["info"]["nket"]]
            lm, mr = self._connection(lk, ld, mk, md), self._connection(mk, md, ak, ad)
            lpos = {q: j for j, q in enumerate(lk)}
            for b in range(len(ln)):
                ql, qr = (int(ln[b]), int(ltw[b]), int(lpg[b])), (int(rn[b]), int(rtw[b]), int(rpg[b]))
                nb, nk = int(kinfo["nbra"][b]), int(kinfo["nket"][b])
                ib = lpos.get(ql)
                if ib is None or qr not in mr:
                    continue
                ti = t["info"]
                if mr[qr][0] != nk or int(ti["nbra"][ib]) != nb:
                    return None, None
                w = np.zeros((ld[ib], nk))
                for (im, ia), off in mr[qr][1].items():
                    cols = md[im] * ad[ia]
                    for qlm in self._fuse(lk[ib], mk[im]):
                        src = blocks.get((qlm, ak[ia]))
                        if src is None:
                            continue
                        if src.shape[0] != lm[qlm][0]:
                            return None, None
                        p = lm[qlm][1][(ib, im)]
                        f = self._recoupling(ak[ia][1], mk[im][1], ttw, lk[ib][1], qr[1], qlm[1])
                        w[:, off:off + cols] += f * src[p:p + ld[ib] * md[im], :].reshape(ld[ib], cols)
                o = t["base"] + int(ti["ntot"][ib])
                lblk = t["data"][o:o + nb * ld[ib]].reshape(nb, ld[ib])
                o = int(kinfo["ntot"][b])
                out[o:o + nb * nk] = (lblk @ w).reshape(-1)
        nrm = np.linalg.norm(out)
        if not np.isfinite(nrm) or nrm < 1e-8:
            return None, None
        return out / nrm, "previous"

    def _bond_labels(self, info, right):
        """(n, 2S or 2Sz, pg) of the bond index of every block of a two-site wavefunction: its right label when the
        right bond is kept, else its left one (src/core/symmetry.hpp:654-731, 1183-1306)"""
        n, tl, tw, pg = _fields(info["q"])
        dn, _, dtw, dpg = _fields(np.array([info["dq"]], np.uint64))
        if self.sym == "sz":
            # SZ: 2Sz is a SIGNED 16-bit field and adds like n; the stored label of a psi block is its (negated) right
            # label, the left one is label + dq
            sgn = lambda t: np.where(t >= 32768, t - 65536, t)
            tw, dtw = sgn(tw), sgn(dtw)
            return (-n, -tw, pg) if right else (n + dn[0], tw + dtw[0], pg ^ dpg[0])
        if right:      # SU2, right label of a psi block: -ket  ->  (-n, twos, pg)
            return -n, tw, pg
        return n + dn[0], tl, pg ^ dpg[0]  # left label: get_bra(dq)  ->  (n + dq.n, twos_low, pg ^ dq.pg)

    def _density_blocks(self, right):
        """the wavefunctions whose density matrices are summed at this bond: psi and, on a noisy sweep, the perturbed
        wavefunctions scaled as MovingEnvironment::scale_perturbative_noise does (src/dmrg/moving_environment.hpp:3655-3671:
        every matrix normalised, then the group scaled to norm sqrt(noise)); density_matrix() adds the matrices 1.. of the
        group, not matrix 0 (:3530-3536).  Returns [(data, info, labels)]"""
        psi, kinfo = self.psi
        out = [(psi, kinfo, self._bond_labels(kinfo, right))]
        if self.pket is not None and self.pket["noise"] != 0:
            pk, tiny = self.pket, 1e-20
            mats = []
            for inf, off in zip(pk["infos"], pk["offs"]):
                m = pk["data"][off:off + inf["len"]].copy()
                nm = np.linalg.norm(m)
                if abs(nm) > tiny:
                    m /= nm
                mats.append(m)
            tot = np.sqrt(sum(float(m @ m) for m in mats))
            if abs(tot) > tiny:
                mats = [m * (np.sqrt(pk["noise"]) / tot) for m in mats]
            for j in range(1, len(mats)):
                out.append((mats[j], pk["infos"][j], self._bond_labels(pk["infos"][j], right)))
        return out

    def split_site(self, forward):
        """the decomposition of the wavefunction just solved, made BEFORE the move to the next site — where block2 makes it
        (DMRG::update_two_dot, sweep_algorithm.hpp:940-960) — from the rotation event the move will consume; at the
        turn-around site of a sweep there is none (the next sweep solves the same two sites again).  Returns
        {"error": discarded weight, "mmps": kept states} or None."""
        n, i = self.n_sites, self.site_key[1]
        if (forward and i == n - 2) or (not forward and i == 0):
            return None
        num, d = self.fx.find_next("lrot" if forward else "rrot")
        if d is None:
            return None
        self.ahead = (num, self._split(d, not forward))
        return self.last_split

    def _mps_tensor(self, d, a):
        """(data, info, offset of block 0, bond labels, bond dimensions) of the MPS tensor a rotation event is made with: a left
        tensor is (fused rows) x (kept states), a right one (kept states) x (fused columns); block label = bond label"""
        info = _info(d, d["mps.info"][0])
        n, _, tw, pg = _fields(info["q"])
        if self.sym == "sz":
            tw = np.where(tw >= 32768, tw - 65536, tw)
        keys = [(int(x), int(y), int(z)) for x, y, z in zip(n, tw, pg)]
        return {"data": np.asarray(a, np.float64), "info": info, "base": int(d["mps.off"][0]), "keys": keys}

    def _take_split(self, d, right):
        num = self.fx.events[self.fx.pos - 1][0]
        if self.ahead is not None and self.ahead[0] == num:
            a, self.ahead = self.ahead[1], None
            return a
        self.ahead = None
        return self._split(d, right)

    def _split(self, d, right):
        """new MPS tensor = the dominant eigenvectors of the density matrix of psi (DensityMatrix decomposition,
        src/dmrg/moving_environment.hpp density_matrix :3512-3538 / split_density_matrix :4218-), per quantum-number sector
        of the bond, kept-state counts as in the fixture's tensor info (the bond dimension bookkeeping is block2's).
        With check_truncation the step also makes block2's choice itself — all eigenvalues of all sectors sorted, the
        largest k kept (truncate_density_matrix, :3674-3800, TruncationTypes::Physical) — and logs, per site, whether its
        per-sector counts equal the fixture's and how far the last kept weight is from the first discarded one."""
        t0 = time.perf_counter()
        ainfo = _info(d, d["mps.info"][0])
        out = np.zeros(int(d["meta"][7]))
        base = int(d["mps.off"][0])
        an, _, atw, apg = _fields(ainfo["q"])
        if self.sym == "sz":
            atw = np.where(atw >= 32768, atw - 65536, atw)
        srcs = self._density_blocks(right)

        # blocks by bond label, in the order (wavefunction, block) the sums are taken in (a noisy sweep adds ~50 perturbed
        # wavefunctions: a label search per (sector, wavefunction) was most of the split's time)
        by_label = {}
        for data, inf, (bn, btw, bpg) in srcs:
            for i, k in enumerate(zip(bn.tolist(), btw.tolist(), bpg.tolist())):
                by_label.setdefault(k, []).append((data, inf, i))

        def rho_of(key, fused):
            rho = np.zeros((fused, fused))
            for data, inf, i in by_label.get(key, ()):
                b = data[inf["ntot"][i]:inf["ntot"][i] + inf["nbra"][i] * inf["nket"][i]].reshape(
                    int(inf["nbra"][i]), int(inf["nket"][i]))
                assert (b.shape[1] if right else b.shape[0]) == fused
                rho += b.T @ b if right else b @ b.T
            return rho

        kept_of, spectrum, rot = {}, {}, {}
        kept_w, mmps = 0.0, 0
        with _one_blas_thread():  # (sector-sized eigenproblems: one LAPACK thread each is the fastest they run)
            for s in range(len(ainfo["q"])):
                key = (int(an[s]), int(atw[s]), int(apg[s]))
                rows, cols = int(ainfo["nbra"][s]), int(ainfo["nket"][s])
                fused, kept = (cols, rows) if right else (rows, cols)
                w, u = np.linalg.eigh(rho_of(key, fused))
                kept_of[key], spectrum[key] = kept, w[::-1]
                kept_w, mmps = kept_w + float(w[::-1][:kept].sum()), mmps + kept
                u = u[:, ::-1][:, :kept]  # largest weights first
                rot[key] = u
                blk = u.T if right else u
                o = base + int(ainfo["ntot"][s])
                out[o:o + rows * cols] = blk.reshape(-1)
        trace = sum(float(data @ data) for data, _, _ in srcs)
        self.last_split = {"error": max(0.0, trace - kept_w), "mmps": mmps}
        if self.use_previous:
            # the other half of the decomposition — block2's new wavefunction tensor, S.V^T of the forward step (rotation^T . psi)
            # or U.S of the backward step (psi . rotation^T) — kept per (left label, right label) for the next site's guess
            psi, kinfo, (bn, btw, bpg) = srcs[0]
            on, otw, opg = self._bond_labels(kinfo, not right)
            blocks = {}
            for i in range(len(bn)):
                u = rot.get((int(bn[i]), int(btw[i]), int(bpg[i])))
                if u is None or u.shape[1] == 0:
                    continue
                b = psi[kinfo["ntot"][i]:kinfo["ntot"][i] + kinfo["nbra"][i] * kinfo["nket"][i]].reshape(
                    int(kinfo["nbra"][i]), int(kinfo["nket"][i]))
                kb, ko = (int(bn[i]), int(btw[i]), int(bpg[i])), (int(on[i]), int(otw[i]), int(opg[i]))
                blocks[(ko, kb) if right else (kb, ko)] = b @ u if right else u.T @ b
            self.carry = {"forward": not right, "site": self.site_key[1], "blocks": blocks}
        if self.check_truncation:
            self._check_truncation(right, kept_of, spectrum, rho_of)
        self.tm.add("split", t0)
        return out

    def _check_truncation(self, right, kept_of, spectrum, rho_of):
        """this loop's own global choice of the kept states against the bond dimensions of the fixture (see _split)"""
        psi, kinfo = self.psi
        bn, btw, bpg = self._bond_labels(kinfo, right)
        fdim = kinfo["nket"] if right else kinfo["nbra"]
        for i in range(len(bn)):  # sectors of the density matrix the reference kept nothing of
            key = (int(bn[i]), int(btw[i]), int(bpg[i]))
            if key not in spectrum:
                spectrum[key] = np.linalg.eigvalsh(rho_of(key, int(fdim[i])))[::-1]
                kept_of[key] = 0
        keys = sorted(spectrum)
        allw = np.concatenate([spectrum[k] for k in keys])
        owner = np.concatenate([np.full(len(spectrum[k]), j) for j, k in enumerate(keys)])
        k_tot = int(sum(kept_of.values()))
        order = np.argsort(-allw, kind="stable")
        mine = np.bincount(owner[order[:k_tot]], minlength=len(keys))
        theirs = np.array([kept_of[k] for k in keys])
        last_kept = float(allw[order[k_tot - 1]]) if k_tot else 0.0
        first_out = float(allw[order[k_tot]]) if k_tot < len(allw) else 0.0
        rec = {"k": k_tot, "n_states": int(len(allw)), "same_counts": bool((mine == theirs).all()),
               "last_kept": last_kept, "first_discarded": first_out, "w_max": float(allw.max()),
               "discarded_weight": float(np.clip(allw[order[k_tot:]], 0, None).sum()), "mismatch": []}
        if not rec["same_counts"]:
            # every state on which the two choices differ lies between the smallest weight the FIXTURE keeps and the largest
            # it discards; for a tie (a degenerate band at the cut) that interval is tiny relative to w_max
            lo, hi = np.inf, -np.inf
            for j, kk in enumerate(keys):
                w = spectrum[kk]
                if theirs[j] != mine[j]:
                    a, b = sorted((int(theirs[j]), int(mine[j])))
                    lo, hi = min(lo, float(w[a:b].min())), max(hi, float(w[a:b].max()))
                    rec["mismatch"].append((kk, int(theirs[j]), int(mine[j])))
            rec["band"] = (lo, hi)
            rec["band_rel_width"] = (hi - lo) / rec["w_max"]
        ref = self.fx.ref_spectra.get(self.site_key)
        if ref is not None:  # the reference's spectrum of this bond: singular values of every sector before the cut
            rw = np.sort(ref[1] ** 2)[::-1]
            mw = np.sort(np.clip(allw, 0, None))[::-1]
            m = min(len(rw), len(mw))
            rec["ref_discarded_weight"] = ref[0]
            rec["spectrum_max_abs_diff"] = float(np.abs(rw[:m] - mw[:m]).max()) if m else 0.0
            rec["n_states_ref"] = int(len(rw))
        self.trunc_log[self.site_key] = rec

    # ---- the loop ---------------------------------------------------------------------------------------------
    def init_environments(self):
        """MovingEnvironment::init_environments (moving_environment.hpp:1245-): all right blocks of the starting state,
        from the last site inwards, with the MPS tensors of the starting state (data of the fixture)"""
        fx = self.fx
        n_rot = sum(1 for n, k, _ in fx.events if k == "rrot" and fx.meta[n][1] < 0)
        self.n_sites = n_rot + 2  # right blocks R[n-1] ... R[2] (the first two sites form the first wavefunction)
        j = self.n_sites - 1
        enl = self._assign(fx.next("rasg")[1])
        while True:
            d = fx.next("rrot")[1]
            self.mpsR[j] = self._mps_tensor(d, d["arena"])
            self.R[j] = self._rotate_and_transform(d, enl, d["arena"])
            enl.close()
            if j == 2:
                break
            enl = self._block(fx.next("rblk")[1], self.R[j])
            j -= 1
            self.basis[j] = self._last_basis
        return sorted(self.R)

    def sweep(self, isw, forward):
        """DMRG::sweep (sweep_algorithm.hpp:2550-2699) with update_two_dot per site; returns the site energies"""
        fx, n = self.fx, self.n_sites
        sites = range(0, n - 1) if forward else range(n - 2, -1, -1)
        out = []
        for i in sites:
            t_site = time.perf_counter()
            self._move_to(i, forward)
            d = self._eham_event(isw, i)
            dn = None
            if fx.peek() == "enoise":  # a noisy sweep of the reference: the perturbative-noise step follows the solve
                dn = fx.next("enoise")[1]
                assert (int(dn["chain.meta"][0]), int(dn["chain.meta"][1])) == (isw, i)
            e, ndav, psi, kinfo, n_pairs = self._eigs(d, dn)
            self._finish_site(isw, i, e, ndav, psi, kinfo)
            out.append(e)
            self.tm.add("site_total", t_site)
        return out

    def _eham_event(self, isw, i):
        _, d = self.fx.next("eham")
        assert (int(d["chain.meta"][0]), int(d["chain.meta"][1])) == (isw, i)
        self._eham_num = self.fx.events[self.fx.pos - 1][0]
        return d

    def _finish_site(self, isw, i, e, ndav, psi, kinfo):
        self.psi = (psi, kinfo)
        self.site_key = (isw, i)
        self.energies[(isw, i)], self.ndav[(isw, i)] = e, ndav
        self.guess_log[(isw, i)] = getattr(self, "_guess_how", None)
        self.carry = None  # (set again by the split of this site; none follows at the turn-around of a sweep)

    def _move_to(self, i, forward):
        """MovingEnvironment::move_to(i) + the two blockings of the site: the enlarged left / right blocks of site i in HBM"""
        fx, n = self.fx, self.n_sites
        self._at = (i, forward)
        if forward:
            if i > 0:  # move_to(i): rotate the enlarged left block of the previous site with the new MPS tensor
                _, d = fx.next("lasg", "lblk")   # (the reference re-contracts it; it is still in HBM here)
                _, d = fx.next("lrot")
                a = self._take_split(d, False)
                self.mpsL[i] = self._mps_tensor(d, a)  # tensor of site i - 1; its column bond is left_dims[i]
                self.L[i] = self._rotate_and_transform(d, self.EL, a)
            if self.EL is not None:
                self.EL.close()
            if i == 0:
                self.EL = self._assign(fx.next("lasg")[1])
            else:
                self.EL = self._block(fx.next("lblk")[1], self.L[i])
                self.basis[i] = self._last_basis
            if self.ER is not None:
                self.ER.close()
            if i == n - 2:
                self.ER = self._assign(fx.next("rasg")[1])
            else:
                self.ER = self._block(fx.next("rblk")[1], self.R[i + 2])
                self.basis[i + 1] = self._last_basis
        else:
            if i < n - 2:  # move_to(i): rotate the enlarged right block of the previous site
                _, d = fx.next("rasg", "rblk")
                _, d = fx.next("rrot")
                a = self._take_split(d, True)
                self.mpsR[i + 2] = self._mps_tensor(d, a)  # tensor of site i + 2; its row bond is right_dims[i + 2]
                if i + 2 in self.R:
                    self.R[i + 2].close()
                self.R[i + 2] = self._rotate_and_transform(d, self.ER, a)
            if self.ER is not None:
                self.ER.close()
            if i == n - 2:
                self.ER = self._assign(fx.next("rasg")[1])
            else:
                self.ER = self._block(fx.next("rblk")[1], self.R[i + 2])
                self.basis[i + 1] = self._last_basis
            if self.EL is not None:
                self.EL.close()
            if i == 0:
                self.EL = self._assign(fx.next("lasg")[1])
            else:
                self.EL = self._block(fx.next("lblk")[1], self.L[i])
                self.basis[i] = self._last_basis


class SumMPODMRG:
    """The sum-MPO calculation H = sum_r H_r (ParallelRuleSimple partition of the integrals, one MPO / set of environments /
    plan per rank, src/dmrg/parallel_simple.hpp:56-99, src/core/parallel_tensor_functions.hpp:51-55) carried through whole
    sweeps in ONE process: every rank's environments are moved with that rank's events, every site is solved ONCE over the
    sum of the ranks' plans (on one GPU per rank that sum is the all-reduce of sigma; here the plans accumulate into the same
    device vector), and every rank rotates its blocks with the same new MPS tensor.  `fixtures` = the per-rank event
    chains of one reference run under mpirun (oracle/ref_dump.cpp para=ij chain=...).
    Tests: the 2-rank ParallelRuleSimple run of the reference (tests/golden/chain_n2su2_ij) and H = H + H
    (tests/test_sweep_gpu.py)."""

    def __init__(self, fixtures, sym, **kw):
        self.ranks = [DMRG(fx, sym, **kw) for fx in fixtures]
        self.energies = {}

    def init_environments(self):
        out = [r.init_environments() for r in self.ranks]
        self.n_sites = self.ranks[0].n_sites
        assert all(r.n_sites == self.n_sites for r in self.ranks)
        return out[0]

    def sweep(self, isw, forward):
        n = self.n_sites
        sites = range(0, n - 1) if forward else range(n - 2, -1, -1)
        out = []
        for i in sites:
            parts, noise = [], []
            for r in self.ranks:
                r._move_to(i, forward)
                d = r._eham_event(isw, i)
                if r.fx.peek() == "enoise":  # a noisy sweep: every rank recorded its own perturbative-noise step
                    noise.append(r.fx.next("enoise")[1])
                parts.append(r._eff_ham(d))
            for r in self.ranks[1:]:  # (rank 0 queues its own next site inside _solve; the others' helpers start here)
                r._prefetch_next()
            e, ndav, psi, kinfo, _ = self.ranks[0]._solve(parts, noise or None)
            for r in self.ranks:
                r._finish_site(isw, i, e, ndav, psi, kinfo)
                r.pket = self.ranks[0].pket  # (the root's sum; every rank splits with the same density matrix)
            self.energies[(isw, i)] = e
            out.append(e)
        return out
