"""The small device vector operations of the C ABI added for the sweep loop, each against numpy:
b2x_vec_gather (many ranges, one launch), b2x_vec_pair_dots (independent dot products, one host round trip),
b2x_vec_olsen_prepare_to (the first half of olsen_precondition, iterative_matrix_functions.hpp:93-108, out of place)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gather_many_ranges(gpu):
    rng = np.random.default_rng(7)
    src = rng.standard_normal(300000)
    lens = np.concatenate([rng.integers(1, 40, 500), [0, 1, 32768, 32769, 70001]]).astype(np.uint64)  # incl. empty, piece edges
    so = rng.integers(0, len(src) - int(lens.max()), len(lens)).astype(np.uint64)
    do = np.concatenate([[0], np.cumsum(lens + 3)[:-1]]).astype(np.uint64)  # disjoint in dst, with gaps
    total = int(do[-1] + lens[-1]) + 5
    d_src, d_dst = gpu.DeviceBuffer(len(src), src), gpu.DeviceBuffer(total, np.full(total, -7.0))
    gpu.gather_d2d(d_dst.ptr, d_src.ptr, do, so, lens)
    got = d_dst.download()
    want = np.full(total, -7.0)
    for a, b, n in zip(do, so, lens):
        want[int(a):int(a + n)] = src[int(b):int(b + n)]
    assert np.array_equal(got, want)
    gpu.gather_d2d(d_dst.ptr, d_src.ptr, [], [], [])  # nothing to do


def test_pair_dots_and_olsen_to(gpu):
    rng = np.random.default_rng(8)
    n = 100003
    vs = [rng.standard_normal(n) for _ in range(9)]
    bufs = [gpu.DeviceBuffer(n, v) for v in vs]
    pairs = [(i, j) for i in range(9) for j in range(i, 9)] + [(0, 0)] * 60  # 105 pairs (<= 128)
    us = (C.c_void_p * len(pairs))(*[bufs[i].ptr for i, _ in pairs])
    ws = (C.c_void_p * len(pairs))(*[bufs[j].ptr for _, j in pairs])
    out = np.zeros(len(pairs))
    gpu.check(gpu.lib().b2x_vec_pair_dots(us, ws, C.c_int(len(pairs)), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), None))
    want = np.array([vs[i] @ vs[j] for i, j in pairs])
    assert np.abs(out - want).max() <= 1e-10 * max(1.0, np.abs(want).max())
    assert gpu.lib().b2x_vec_pair_dots(us, ws, C.c_int(129), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), None) != 0
    # olsen, out of place: q stays, q_out = q / (ld - diag), t = c / (ld - diag) where |ld - diag| > 1e-12
    q, c, diag = vs[0], vs[1], rng.uniform(-1, 1, n)
    ld = float(diag[5])  # one exactly singular element
    dq, dc, dd = gpu.DeviceBuffer(n, q), gpu.DeviceBuffer(n, c), gpu.DeviceBuffer(n, diag)
    dqo, dt = gpu.DeviceBuffer(n), gpu.DeviceBuffer(n)
    gpu.check(gpu.lib().b2x_vec_olsen_prepare_to(C.c_void_p(dq.ptr), C.c_void_p(dqo.ptr), C.c_void_p(dt.ptr), C.c_void_p(dc.ptr),
                                                 C.c_void_p(dd.ptr), C.c_double(ld), C.c_size_t(n), None))
    gpu.device_sync()
    den = ld - diag
    ok = np.abs(den) > 1e-12
    assert np.array_equal(dq.download(), q)
    assert np.allclose(dqo.download(), np.where(ok, q / np.where(ok, den, 1.0), q), rtol=1e-14, atol=0)
    assert np.allclose(dt.download(), np.where(ok, c / np.where(ok, den, 1.0), c), rtol=1e-14, atol=0)


def test_gs_finish_on_device(gpu):
    """b2x_vec_gs_finish: second Gram-Schmidt pass + normalisation with the coefficients left on the device, against numpy;
    the degenerate case (v in the span of the basis) raises the status flag instead of dividing by a rounding-sized norm"""
    rng = np.random.default_rng(9)
    n, m = 50021, 7
    q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    v = rng.standard_normal(n)
    bufs = [gpu.DeviceBuffer(n, np.ascontiguousarray(q[:, j])) for j in range(m)]
    bs = (C.c_void_p * m)(*[b.ptr for b in bufs])
    dv, dout = gpu.DeviceBuffer(n, v), gpu.DeviceBuffer(n)
    flag = C.c_int(-1)
    gpu.check(gpu.lib().b2x_vec_gs_finish(bs, C.c_int(m), C.c_void_p(dv.ptr), C.c_void_p(dout.ptr), C.c_size_t(n), None))
    gpu.device_sync()
    gpu.check(gpu.lib().b2x_vec_gs_status(C.byref(flag), C.c_int(1)))
    want = v - q @ (q.T @ v)
    want /= np.linalg.norm(want)
    got = dout.download()
    assert flag.value == 0 and np.abs(got - want).max() < 1e-12 and abs(np.linalg.norm(got) - 1) < 1e-12
    # m = 0: plain normalisation
    gpu.check(gpu.lib().b2x_vec_gs_finish(None, C.c_int(0), C.c_void_p(dv.ptr), C.c_void_p(dout.ptr), C.c_size_t(n), None))
    gpu.device_sync()
    assert np.abs(dout.download() - v / np.linalg.norm(v)).max() < 1e-13
    # degenerate: v = a combination of the basis
    dv.upload(q @ rng.standard_normal(m))
    gpu.check(gpu.lib().b2x_vec_gs_finish(bs, C.c_int(m), C.c_void_p(dv.ptr), C.c_void_p(dout.ptr), C.c_size_t(n), None))
    gpu.device_sync()
    gpu.check(gpu.lib().b2x_vec_gs_status(C.byref(flag), C.c_int(1)))
    assert flag.value == 1 and np.isfinite(dout.download()).all()
    gpu.check(gpu.lib().b2x_vec_gs_status(C.byref(flag), C.c_int(0)))
    assert flag.value == 0  # cleared by the reset above


def test_ritz_olsen_one_pass(gpu):
    """b2x_vec_ritz_olsen against numpy: x = sum a_j b_j, q = sum a_j s_j - theta x, q2 = q / (theta - diag), t = x / (theta - diag)"""
    rng = np.random.default_rng(10)
    n, m = 70001, 11
    b, s_ = rng.standard_normal((m, n)), rng.standard_normal((m, n))
    a, theta, diag = rng.standard_normal(m), 0.37, rng.uniform(-1, 1, n)
    diag[3] = theta  # a singular element: left unscaled
    db, ds = [gpu.DeviceBuffer(n, np.ascontiguousarray(v)) for v in b], [gpu.DeviceBuffer(n, np.ascontiguousarray(v)) for v in s_]
    pb, ps = (C.c_void_p * m)(*[v.ptr for v in db]), (C.c_void_p * m)(*[v.ptr for v in ds])
    dd = gpu.DeviceBuffer(n, diag)
    out = [gpu.DeviceBuffer(n) for _ in range(4)]
    gpu.check(gpu.lib().b2x_vec_ritz_olsen(pb, ps, C.c_int(m), a.ctypes.data_as(C.c_void_p), C.c_double(theta), C.c_void_p(dd.ptr),
                                           *[C.c_void_p(o.ptr) for o in out], C.c_size_t(n), None))
    gpu.device_sync()
    x = a @ b
    q = a @ s_ - theta * x
    d = theta - diag
    ok = np.abs(d) > 1e-12
    want = [x, q, np.where(ok, q / np.where(ok, d, 1.0), q), np.where(ok, x / np.where(ok, d, 1.0), x)]
    for o, w in zip(out, want):
        assert np.abs(o.download() - w).max() <= 1e-12 * max(1.0, np.abs(w).max())
