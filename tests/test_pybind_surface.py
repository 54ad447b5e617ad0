"""The pybind names of the hot-path classes SURVEY §8(b) lists (src/pybind/pybind_core.hpp:1100-1340, 1726-1731;
pybind_dmrg.hpp:1140-): Threading.seq_type / Global.threading, OperatorFunctions, TensorFunctions,
ParallelTensorFunctions, ParallelCommunicator (+ the RCCL one), EffectiveKernel, DavidsonTypes.  CPU part: the objects exist
and behave without a device; the GPU part runs them on the reference's fixtures."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from block2_preview_amd.planfile import PAIR_DTYPE, read_arrays, read_plan


def _sym(fn):
    return "su2" if "su2" in os.path.basename(fn) else "sz"


def test_names_exist_and_defaults(built):
    from block2_preview_amd import b2x_host as h

    assert h.Global.threading.seq_type == h.SeqTypes.Device
    for sub in (h.su2, h.sz):
        opf = sub.OperatorFunctions()
        assert opf.seq.mode == h.SeqTypes.Device and opf.seq.n_pairs == 0
        tf = sub.TensorFunctions(opf)
        assert tf.opf is opf and tf.comm is None
    # OperatorFunctions takes the mode of its sequence from the global threading scheme (operator_functions.hpp:73-75)
    old = h.Global.threading.seq_type
    try:
        h.Global.threading.seq_type = h.SeqTypes.Auto
        assert h.su2.OperatorFunctions().seq.mode == h.SeqTypes.Auto
        t = h.Threading()
        t.seq_type = h.SeqTypes.Tasked
        h.Global.threading = t
        assert h.sz.OperatorFunctions().seq.mode == h.SeqTypes.Tasked
    finally:
        t0 = h.Threading()
        t0.seq_type = old
        h.Global.threading = t0
    # the serial communicator: size / rank / root, collectives must not be called (parallel_rule.hpp:56-307)
    c = h.ParallelCommunicator()
    assert (c.size, c.rank, c.root, c.is_root(), c.tcomm) == (1, 0, 0, True, 0.0)
    for call in (lambda: c.allreduce_sum(0, 1), lambda: c.broadcast(0, 1, 0), c.barrier):
        with pytest.raises(RuntimeError):
            call()
    assert int(h.DavidsonTypes.HarmonicCloseTo) == 20 and int(h.DavidsonTypes.NoPrecond) == 64
    assert isinstance(h.EffectiveKernel(), h.EffectiveKernel)


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(GOLDEN, "rot_*.erot"))), ids=os.path.basename)
def test_tensor_functions_rotate_records_reference_pairs(built, fn):
    """TensorFunctions.left_rotate / right_rotate == the pairs the reference's tensor_rotate recorded; the wrong
    direction is refused"""
    from block2_preview_amd import b2x_host as h

    d = read_arrays(fn)
    sub = getattr(h, _sym(fn))
    tf = sub.TensorFunctions(sub.OperatorFunctions())
    right = int(d["meta"][2]) != 0
    pairs_b, _ = (tf.right_rotate if right else tf.left_rotate)(d)
    mine = np.frombuffer(bytes(pairs_b), PAIR_DTYPE)
    ref = read_plan(fn.replace(".erot", ".plan"))
    for name in PAIR_DTYPE.names:
        assert np.array_equal(mine[name], ref.pairs[name]), name
    with pytest.raises(RuntimeError):
        (tf.left_rotate if right else tf.right_rotate)(d)


@pytest.mark.gpu
def test_tensor_functions_call_and_parallel_call(gpu, tmp_path):
    """TensorFunctions::operator() (tensor_functions.hpp:59-62) and ParallelTensorFunctions::operator()
    (parallel_tensor_functions.hpp:51-55; one-rank RCCL communicator) give the reference's sigma"""
    from block2_preview_amd import b2x_host as h

    h.device_init(0)
    pf = read_plan(os.path.join(GOLDEN, "n2su2.sw0.site4.plan"))
    opf = h.su2.OperatorFunctions()
    opf.seq.load_pairs(pf.pairs, pf.arena)
    tf = h.su2.TensorFunctions(opf)
    sig = np.zeros(pf.sigma_len)
    tf(pf.psi, sig, 1.0)
    assert np.abs(sig - pf.sigma_ref).max() <= 1e-12 * np.abs(pf.sigma_ref).max()
    comm = h.RCCLCommunicator(0, 1, str(tmp_path / "id"))
    ptf = h.su2.ParallelTensorFunctions(opf, comm)
    sig2 = np.zeros(pf.sigma_len)
    ptf(pf.psi, sig2, 1.0)
    assert np.array_equal(sig2, sig) and comm.tcomm > 0 and comm.handle != 0 and ptf.comm is comm
    opf.seq.clear()


@pytest.mark.gpu
def test_effective_kernel_hook(gpu):
    """eff_kernel.compute(beta, f, a, b, xs) wraps every matrix-vector product of eigs (effective_hamiltonian.hpp:515-519);
    a Python override that forwards to f sees Ndav calls and leaves the result unchanged; one that shifts H by a constant
    (b += beta (H + 3) a, through the C ABI's axpy on the device vectors) shifts the eigenvalue by 3"""
    import ctypes as C

    from block2_preview_amd import b2x_host as h

    h.device_init(0)
    pf = read_plan(os.path.join(GOLDEN, "n2su2.sw2.site5.plan"))

    def make():
        seq = h.BatchGEMMSeq()
        seq.load_pairs(pf.pairs, pf.arena)
        return h.EffectiveHamiltonian(seq, pf.diag.tolist())

    eh = make()
    e0, nd0, _, _, ket0 = eh.eigs(pf.psi.tolist(), conv_thrd=1e-12, max_iter=500)
    eh.post_precompute()

    class Count(h.EffectiveKernel):
        calls = 0

        def compute(self, beta, f, a, b, xs):
            Count.calls += 1
            assert xs == [] and beta == 1.0
            f(a, b, beta)

    eh = make()
    eh.eff_kernel = Count()
    e1, nd1, _, _, ket1 = eh.eigs(pf.psi.tolist(), conv_thrd=1e-12, max_iter=500)
    eh.post_precompute()
    assert (e1, nd1) == (e0, nd0) and Count.calls == nd0 and np.array_equal(ket1, ket0)

    n = pf.psi_len

    class Shift(h.EffectiveKernel):
        def compute(self, beta, f, a, b, xs):
            f(a, b, beta)
            gpu.check(gpu.lib().b2x_vec_axpy(C.c_double(3.0 * beta), C.c_void_p(a), C.c_void_p(b), C.c_size_t(n), None))

    eh = make()
    eh.eff_kernel = Shift()
    e2, _, _, _, _ = eh.eigs(pf.psi.tolist(), conv_thrd=1e-12, max_iter=500)
    eh.post_precompute()
    assert abs(e2 - (e0 + 3.0)) < 1e-9
