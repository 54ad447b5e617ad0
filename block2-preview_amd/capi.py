"""ctypes binding of the C ABI in include/b2x.h (libb2x.so, built in-tree by __graft_entry__.build()).

This is the reference-side binding a Python host would use; the pybind/C++ host of block2 would bind
the same symbols.  There is deliberately no fallback: if libb2x.so is missing, or no
gfx950 device is present, the compute entry points raise.
"""
import ctypes as C
import os

import numpy as np

from .planfile import GEMM_DTYPE, OUTER_TERM_DTYPE, PAIR_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("B2X_LIB") or os.path.join(_HERE, "libb2x.so")  # B2X_LIB: an experimental build (kernel probes)

# every symbol include/b2x.h declares (checked by tests/test_capi_symbols.py)
DECLARED_SYMBOLS = [
    "b2x_last_error", "b2x_version", "b2x_device_count", "b2x_device_init", "b2x_device_sync",
    "b2x_device_alloc", "b2x_device_free", "b2x_memcpy_h2d", "b2x_memcpy_d2h",
    "b2x_arena_create", "b2x_arena_adopt_device", "b2x_arena_resolve", "b2x_arena_len",
    "b2x_arena_device_ptr", "b2x_arena_destroy",
    "b2x_plan_create", "b2x_plan_execute", "b2x_plan_get_stats", "b2x_plan_time_kernel", "b2x_plan_destroy",
    "b2x_plan_cache_stats", "b2x_plan_cache_clear", "b2x_trim",
    "b2x_gemm_plan_create", "b2x_outer_build",
    "b2x_vec_dot", "b2x_vec_axpy", "b2x_vec_scal", "b2x_vec_copy", "b2x_vec_zero", "b2x_vec_precondition",
    "b2x_vec_multi_dot", "b2x_vec_pair_dots", "b2x_vec_olsen_prepare_to", "b2x_vec_gather", "b2x_vec_gs_finish", "b2x_vec_ritz_olsen", "b2x_vec_gs_status", "b2x_outer_plan_create", "b2x_outer_plan_execute", "b2x_outer_plan_destroy", "b2x_vec_lincomb", "b2x_vec_olsen_prepare", "b2x_diag_build",
    "b2x_comm_init", "b2x_comm_init_session", "b2x_comm_unique_id", "b2x_comm_init_id", "b2x_comm_rank", "b2x_allreduce_sum", "b2x_broadcast",
    "b2x_barrier", "b2x_comm_destroy",
]


class PlanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_pairs", "macs", "op_elems_unique", "psi_len", "sigma_len", "n_targets", "n_tiles", "n_items",
        "n_parts", "device_bytes", "macs_executed", "dominant_class", "macs_dominant",
        "macs_alg_dominant", "n_launches", "macs_issued", "fallback", "n_staged", "n_flipped", "n_shared_products",
        "n_merged_groups", "n_merged_members")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PlanOptions(C.Structure):
    _fields_ = [("tile_m", C.c_int32), ("tile_n", C.c_int32), ("kernel", C.c_int32), ("_pad", C.c_int32),
                ("item_macs", C.c_int64), ("two_stage", C.c_int32), ("scratch_mb", C.c_int32),
                ("keep_order", C.c_int32), ("presum", C.c_int32), ("reserved", C.c_int32 * 4)]


class B2XError(RuntimeError):
    """Non-zero return from the C ABI (the reference throws runtime_error at this boundary)."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise B2XError("libb2x.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.b2x_last_error.restype = C.c_char_p
        _lib.b2x_version.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise B2XError("b2x error %d: %s" % (rc, lib().b2x_last_error().decode()))


def _ptr(a):
    """device pointer (int) or numpy array -> c_void_p"""
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


def device_count():
    n = C.c_int(0)
    check(lib().b2x_device_count(C.byref(n)))
    return n.value


_device_ordinal = 0


def device_init(ordinal=0):
    """selects the device for the CALLING thread (hipSetDevice is per thread: a helper thread calls this with
    current_device() before its first device call)"""
    global _device_ordinal
    check(lib().b2x_device_init(C.c_int(ordinal)))
    _device_ordinal = int(ordinal)


def current_device():
    return _device_ordinal


def device_sync():
    check(lib().b2x_device_sync())


class DeviceBuffer:
    """fp64 device vector from b2x_device_alloc (exactly n elements)"""

    def __init__(self, n, host=None):
        p = C.c_void_p()
        check(lib().b2x_device_alloc(C.byref(p), C.c_size_t(max(1, n) * 8)))
        self.ptr, self.n = p.value, n
        if host is None:  # zero-filled on the device (no host array, no PCIe copy)
            if n:
                check(lib().b2x_vec_zero(C.c_void_p(self.ptr), C.c_size_t(n), None))
        else:
            self.upload(host)

    def upload(self, host):
        host = np.ascontiguousarray(host, np.float64)
        assert host.size == self.n
        if self.n:
            check(lib().b2x_memcpy_h2d(C.c_void_p(self.ptr), _ptr(host), C.c_size_t(self.n * 8)))

    def download(self):
        out = np.empty(self.n)
        if self.n:
            check(lib().b2x_memcpy_d2h(_ptr(out), C.c_void_p(self.ptr), C.c_size_t(self.n * 8)))
        return out

    def close(self):
        if self.ptr:
            lib().b2x_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Arena:
    """Operator blocks resident in HBM (OperatorTensor::ops[*]->data of the reference)."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    @classmethod
    def from_host(cls, arrays):
        arrays = [np.ascontiguousarray(a, np.float64) for a in arrays]
        n = len(arrays)
        bases = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
        lens = (C.c_size_t * n)(*[a.size for a in arrays])
        h = C.c_void_p()
        check(lib().b2x_arena_create(C.byref(h), C.c_size_t(n), bases, lens))
        return cls(h, arrays)

    @classmethod
    def adopt_device(cls, dev_ptr, length, keep=None):
        h = C.c_void_p()
        check(lib().b2x_arena_adopt_device(C.byref(h), C.c_void_p(int(dev_ptr)), C.c_size_t(length)))
        return cls(h, keep)

    def __len__(self):
        n = C.c_uint64()
        check(lib().b2x_arena_len(self._h, C.byref(n)))
        return n.value

    def resolve(self, host_address):
        off = C.c_uint64()
        check(lib().b2x_arena_resolve(self._h, C.c_void_p(int(host_address)), C.byref(off)))
        return off.value

    def close(self):
        if self._h is not None:
            lib().b2x_arena_destroy(self._h)
            self._h = None
        self._keep = None  # (an adopted device buffer is the caller's again)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Device-resident GEMM-pair plan (BatchGEMMSeq after precompute(), before post_precompute())."""

    def __init__(self, arena, pairs, psi_len, sigma_len, kernel=0, tile_n=0, item_macs=0, two_stage=0, scratch_mb=0,
                 tile_m=0, keep_order=0, presum=0):
        pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
        opt = PlanOptions()
        opt.kernel, opt.tile_n, opt.item_macs = kernel, tile_n, item_macs
        opt.two_stage, opt.scratch_mb, opt.tile_m, opt.keep_order = two_stage, scratch_mb, tile_m, keep_order
        opt.presum = presum
        h = C.c_void_p()
        check(lib().b2x_plan_create(C.byref(h), arena._h, C.c_size_t(len(pairs)), _ptr(pairs), C.c_size_t(psi_len),
                                    C.c_size_t(sigma_len), C.byref(opt)))
        self._h, self._arena = h, arena
        self.psi_len, self.sigma_len = psi_len, sigma_len

    def execute_host(self, psi, sigma, scale=1.0):
        """sigma += scale * H psi on host numpy arrays (copied through the device)."""
        assert psi.dtype == np.float64 and sigma.dtype == np.float64
        assert psi.size == self.psi_len and sigma.size == self.sigma_len
        check(lib().b2x_plan_execute(self._h, _ptr(psi), _ptr(sigma), C.c_double(scale), C.c_int(0), None))

    def execute_device(self, psi_ptr, sigma_ptr, scale=1.0, stream=0):
        check(lib().b2x_plan_execute(self._h, C.c_void_p(int(psi_ptr)), C.c_void_p(int(sigma_ptr)), C.c_double(scale),
                                     C.c_int(1), C.c_void_p(int(stream))))

    def time_kernel(self, psi_ptr, sigma_ptr, n=3, stream=0):
        a, b = C.c_double(), C.c_double()
        check(lib().b2x_plan_time_kernel(self._h, C.c_void_p(int(psi_ptr)), C.c_void_p(int(sigma_ptr)), C.c_int(n),
                                         C.c_void_p(int(stream)), C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def stats(self):
        st = PlanStats()
        check(lib().b2x_plan_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h is not None:
            lib().b2x_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GemmPlan(Plan):
    """Device-resident single-GEMM list (batch[1] of BatchGEMMSeq after a partial multiply, before auto_perform(v)):
    out += scale * sum_i alpha_i op(A_i) op(B_i), operands from the arena or the input vector."""

    def __init__(self, arena, gemms, in_len, out_len, item_macs=0, keep_order=0):
        gemms = np.ascontiguousarray(gemms, GEMM_DTYPE)
        opt = PlanOptions()
        opt.item_macs, opt.keep_order = item_macs, keep_order
        h = C.c_void_p()
        check(lib().b2x_gemm_plan_create(C.byref(h), arena._h, C.c_size_t(len(gemms)), _ptr(gemms), C.c_size_t(in_len),
                                         C.c_size_t(out_len), C.byref(opt)))
        self._h, self._arena = h, arena
        self.psi_len, self.sigma_len = in_len, out_len


def diag_build(arena, terms, diag, diag_len=None, on_device=False, stream=0):
    """diag += sum of rank-1 diagonal terms (b2x_diag_term records as bytes or a structured array); `diag` is a host
    numpy array or, with on_device, a device address of diag_len doubles"""
    terms = np.ascontiguousarray(terms)
    n_terms = terms.nbytes // 56
    if not on_device:
        diag_len = diag.size
    check(lib().b2x_diag_build(arena._h, C.c_size_t(n_terms), _ptr(terms), C.c_size_t(diag_len), _ptr(diag),
                               C.c_int(1 if on_device else 0), C.c_void_p(int(stream))))


def memcpy_d2d(dst_ptr, src_ptr, n_elems):
    """device-to-device copy of n_elems doubles (default stream)"""
    check(lib().b2x_vec_copy(C.c_void_p(int(src_ptr)), C.c_void_p(int(dst_ptr)), C.c_size_t(n_elems), None))


def gather_d2d(dst_ptr, src_ptr, dst_offs, src_offs, lens):
    """dst[dst_offs[i] : + lens[i]] = src[src_offs[i] : + lens[i]] for all i in one launch (element offsets into two device
    vectors; default stream; returns when done)"""
    n = len(lens)
    if n == 0:
        return
    do, so, ln = (np.ascontiguousarray(a, np.uint64) for a in (dst_offs, src_offs, lens))
    check(lib().b2x_vec_gather(C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), C.c_size_t(n), _ptr(do), _ptr(so), _ptr(ln), None))


def outer_build(arena, terms, vin, vout, on_device=False, in_len=None, out_len=None, stream=0):
    """vout += sum of element-wise block-product terms (blocking: a (x) site operator, operator sums).  Host numpy arrays
    (copied through the device) or, with on_device, device pointers + explicit lengths."""
    terms = np.ascontiguousarray(terms, OUTER_TERM_DTYPE)
    if not on_device:
        assert vin.dtype == np.float64 and vout.dtype == np.float64
        in_len, out_len = vin.size, vout.size
    check(lib().b2x_outer_build(arena._h, C.c_size_t(len(terms)), _ptr(terms), _ptr(vin), C.c_size_t(in_len),
                                C.c_size_t(out_len), _ptr(vout), C.c_int(1 if on_device else 0), C.c_void_p(int(stream))))


class OuterPlan:
    """an element-wise term list compiled and uploaded once (b2x_outer_plan_create), executed on device vectors later"""

    def __init__(self, terms, arena_len, in_len, out_len):
        terms = np.ascontiguousarray(terms, OUTER_TERM_DTYPE)
        self._h = C.c_void_p()
        check(lib().b2x_outer_plan_create(C.byref(self._h), C.c_uint64(int(arena_len)), C.c_size_t(len(terms)), _ptr(terms),
                                          C.c_size_t(int(in_len)), C.c_size_t(int(out_len))))

    def execute(self, arena, in_ptr, out_ptr, stream=0):
        check(lib().b2x_outer_plan_execute(self._h, arena._h, C.c_void_p(int(in_ptr)), C.c_void_p(int(out_ptr)),
                                           C.c_void_p(int(stream))))

    def close(self):
        if self._h:
            lib().b2x_outer_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_cache_stats():
    """(hits, misses, cached plans, bytes of cached work lists) of the compiled-plan cache (b2x_plan_cache_stats)"""
    v = [C.c_uint64() for _ in range(4)]
    check(lib().b2x_plan_cache_stats(*[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def plan_cache_clear():
    check(lib().b2x_plan_cache_clear())


def trim():
    """give the library's idle device memory (plan cache + buffer pool) back to the driver; returns the bytes released"""
    v = C.c_uint64()
    check(lib().b2x_trim(C.byref(v)))
    return v.value


class Comm:
    """RCCL communicator of the sum-MPO path (ParallelCommunicator<S> of the reference: allreduce_sum / broadcast /
    barrier on device-resident fp64 vectors).  One process per GPU; call device_init() first."""

    def __init__(self, rank, size, id_file=None, id_bytes=None, nonce=0):
        h = C.c_void_p()
        if id_bytes is not None:
            buf = C.create_string_buffer(bytes(id_bytes), 128)
            check(lib().b2x_comm_init_id(C.byref(h), C.c_int(rank), C.c_int(size), buf))
        elif nonce:
            check(lib().b2x_comm_init_session(C.byref(h), C.c_int(rank), C.c_int(size),
                                              None if id_file is None else os.fsencode(id_file), C.c_uint64(int(nonce))))
        else:
            check(lib().b2x_comm_init(C.byref(h), C.c_int(rank), C.c_int(size),
                                      None if id_file is None else os.fsencode(id_file)))
        self._h, self.rank, self.size = h, rank, size

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().b2x_comm_unique_id(buf))
        return buf.raw

    def rank_size(self):
        """(rank, size) as the communicator itself reports them (b2x_comm_rank)"""
        r, n = C.c_int(-1), C.c_int(-1)
        check(lib().b2x_comm_rank(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def allreduce_sum(self, dev_ptr, n, stream=0):
        check(lib().b2x_allreduce_sum(self._h, C.c_void_p(int(dev_ptr)), C.c_size_t(n), C.c_void_p(int(stream))))

    def broadcast(self, dev_ptr, n, root=0, stream=0):
        check(lib().b2x_broadcast(self._h, C.c_void_p(int(dev_ptr)), C.c_size_t(n), C.c_int(root), C.c_void_p(int(stream))))

    def barrier(self):
        check(lib().b2x_barrier(self._h))

    def close(self):
        if self._h is not None:
            lib().b2x_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
