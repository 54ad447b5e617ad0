import sys, time, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np
from block2_preview_amd import capi
capi.device_init(0)
n=32000
x=capi.DeviceBuffer(n, np.random.rand(n)); y=capi.DeviceBuffer(n, np.random.rand(n)); z=capi.DeviceBuffer(n, np.zeros(n))
L=capi.lib()
r=C.c_double()
def t(f, reps=200):
    f(); capi.device_sync()
    t0=time.perf_counter()
    for _ in range(reps): f()
    capi.device_sync()
    return (time.perf_counter()-t0)/reps*1e6
print("dot (sync)   %.1f us" % t(lambda: L.b2x_vec_dot(C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_size_t(n), C.byref(r), None)))
print("axpy (async) %.1f us" % t(lambda: L.b2x_vec_axpy(C.c_double(0.0), C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_size_t(n), None)))
print("copy (async) %.1f us" % t(lambda: L.b2x_vec_copy(C.c_void_p(x.ptr), C.c_void_p(z.ptr), C.c_size_t(n), None)))
print("zero (async) %.1f us" % t(lambda: L.b2x_vec_zero(C.c_void_p(z.ptr), C.c_size_t(n), None)))
ptrs=(C.c_void_p*8)(*[x.ptr]*8); coef=(C.c_double*8)(*[0.1]*8)
print("lincomb8 (async) %.1f us" % t(lambda: L.b2x_vec_lincomb(ptrs, 8, coef, C.c_void_p(z.ptr), C.c_size_t(n), None)))
res=(C.c_double*8)()
print("multidot8 (sync) %.1f us" % t(lambda: L.b2x_vec_multi_dot(ptrs, 8, C.c_void_p(y.ptr), C.c_size_t(n), res, None)))
def seq():
    L.b2x_vec_axpy(C.c_double(0.0), C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_size_t(n), None)
    L.b2x_vec_dot(C.c_void_p(x.ptr), C.c_void_p(y.ptr), C.c_size_t(n), C.byref(r), None)
print("axpy+dot (sync) %.1f us" % t(seq))
