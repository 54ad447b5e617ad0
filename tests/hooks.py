"""ctypes binding of the TEST-ONLY library tests/native/libb2x_testhooks.so (built by __graft_entry__.build()):
the plan compiler's work lists evaluated with plain host loops.  Not part of the product."""
import ctypes as C
import os

import numpy as np

from block2_preview_amd.capi import B2XError, PlanOptions, PlanStats, _ptr
from block2_preview_amd.planfile import GEMM_DTYPE, OUTER_TERM_DTYPE, PAIR_DTYPE

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libb2x_testhooks.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise B2XError("libb2x_testhooks.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.b2x_test_last_error.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise B2XError("b2x error %d: %s" % (rc, lib().b2x_test_last_error().decode()))


def debug_compile_and_emulate_outer(terms, arena, vin, vout):
    """TEST HOOK: compile an outer-term list into cells / work units and evaluate it with host loops."""
    terms = np.ascontiguousarray(terms, OUTER_TERM_DTYPE)
    nw, ne = C.c_uint64(), C.c_uint64()
    check(lib().b2x_debug_compile_and_emulate_outer(
        C.c_size_t(len(terms)), _ptr(terms), C.c_size_t(vin.size), C.c_size_t(vout.size), C.c_uint64(arena.size),
        _ptr(arena), _ptr(vin), _ptr(vout), C.byref(nw), C.byref(ne)))
    return nw.value, ne.value


def debug_compile_and_emulate_gemms(gemms, in_len, out_len, arena, vin, vout, scale=1.0, item_macs=0, keep_order=0):
    """TEST HOOK: compile a single-GEMM list and evaluate the compiled work list with host loops."""
    gemms = np.ascontiguousarray(gemms, GEMM_DTYPE)
    opt = PlanOptions()
    opt.item_macs, opt.keep_order = item_macs, keep_order
    st = PlanStats()
    check(lib().b2x_debug_compile_and_emulate_gemms(
        C.c_size_t(len(gemms)), _ptr(gemms), C.c_size_t(in_len), C.c_size_t(out_len), C.c_uint64(arena.size),
        _ptr(arena), _ptr(vin), _ptr(vout), C.c_double(scale), C.byref(opt), C.byref(st)))
    return st.as_dict()


def debug_compile_and_emulate(pairs, psi_len, sigma_len, arena, psi, sigma, scale=1.0, tile_n=0, item_macs=0,
                              two_stage=0, scratch_mb=0, keep_order=0, presum=0, arena_len=None):
    """TEST HOOK (not part of include/b2x.h): compile a plan and evaluate the compiled work list with
    host loops, so the plan compiler can be verified without a GPU.  Returns (stats, fallback)."""
    pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
    opt = PlanOptions()
    opt.tile_n, opt.item_macs = tile_n, item_macs
    opt.two_stage, opt.scratch_mb, opt.keep_order, opt.presum = two_stage, scratch_mb, keep_order, presum
    st, fb = PlanStats(), C.c_int(0)
    data = arena is not None and psi is not None and sigma is not None  # else: compile only (structure without data)
    check(lib().b2x_debug_compile_and_emulate(
        C.c_size_t(len(pairs)), _ptr(pairs), C.c_size_t(psi_len), C.c_size_t(sigma_len),
        C.c_uint64(arena.size if arena_len is None else arena_len),
        _ptr(arena) if data else None, _ptr(psi) if data else None, _ptr(sigma) if data else None,
        C.c_double(scale), C.byref(opt), C.byref(st), C.byref(fb)))
    return st.as_dict(), bool(fb.value)
