"""ctypes wrapper of oracle/liboracle.so (hpsi_oracle.c) — TEST INFRASTRUCTURE, see oracle/__init__.py.

Parity pin: hpsi_oracle.c is checked against the golden plans captured from the real reference
(tests/golden/*.plan, produced by oracle/ref_dump.cpp) in tests/test_oracle_golden.py."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    src = os.path.join(_HERE, "hpsi_oracle.c")
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.b2x_oracle_replay.restype = C.c_uint64
        _lib.b2x_oracle_gemm_list.restype = C.c_uint64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def replay(pairs, arena, psi, sigma, scale=1.0, nthreads=1):
    """sigma += scale * H psi (in place); returns MACs.  Restates BatchGEMMSeq::operator() Tasked."""
    pairs = np.ascontiguousarray(pairs)
    assert arena.dtype == np.float64 and psi.dtype == np.float64 and sigma.dtype == np.float64
    return int(lib().b2x_oracle_replay(C.c_uint64(len(pairs)), _p(pairs), _p(arena), _p(psi), _p(sigma),
                                       C.c_uint64(sigma.size), C.c_double(scale), C.c_int(nthreads)))


def dense(pairs, arena, psi_len, sigma_len):
    pairs = np.ascontiguousarray(pairs)
    h = np.zeros((sigma_len, psi_len))
    lib().b2x_oracle_dense(C.c_uint64(len(pairs)), _p(pairs), _p(arena), C.c_uint64(psi_len), C.c_uint64(sigma_len),
                           _p(h))
    return h


def gemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    lib().b2x_oracle_gemm(C.c_int(ta), C.c_int(tb), C.c_int(m), C.c_int(n), C.c_int(k), C.c_double(alpha), _p(a),
                          C.c_int(lda), _p(b), C.c_int(ldb), C.c_double(beta), _p(c), C.c_int(ldc))


def gemm_list(gemms, arena, vin, vout, scale=1.0, nthreads=1):
    """vout += scale * sum of single-GEMM records (in place); returns MACs.  Restates BatchGEMMSeq::auto_perform(v)."""
    gemms = np.ascontiguousarray(gemms)
    assert arena.dtype == np.float64 and vin.dtype == np.float64 and vout.dtype == np.float64
    return int(lib().b2x_oracle_gemm_list(C.c_uint64(len(gemms)), _p(gemms), _p(arena), _p(vin), _p(vout),
                                          C.c_uint64(vout.size), C.c_double(scale), C.c_int(nthreads)))


def outer(terms, arena, vin, vout):
    """vout += element-wise block-product terms (in place).  Restates GMatrixFunctions::tensor_product / iadd."""
    terms = np.ascontiguousarray(terms)
    assert terms.dtype.itemsize == 64 and arena.dtype == np.float64 and vin.dtype == np.float64 and vout.dtype == np.float64
    lib().b2x_oracle_outer(C.c_uint64(len(terms)), _p(terms), _p(arena), _p(vin), _p(vout))
